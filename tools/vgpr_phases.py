"""Development aid: highest VGPR index and scratch instructions of a stage kernel per phase (compiles rmh_api.hip with -DRMH_PHASE_MARKS like
tools/isa_phases.py): where the register pressure of a kernel peaks.

    python tools/vgpr_phases.py <order> <mode> [-D...]        (needs hipcc; no GPU)
"""
import re, subprocess, sys, os
order, mode = int(sys.argv[1]), int(sys.argv[2])
out = "/tmp/rmh_phases_v.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans", "-DRMH_PHASE_MARKS", *sys.argv[3:], "-S", "--cuda-device-only", "remhos_amd/csrc/rmh_api.hip", "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
kern = f"_ZN3rmh10ho_kernel2ILi{order}ELi{mode}EEEvNS_6HoArgsE"
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
phase = -1
mx = {}
cnt = {}
for l in lines[start:end + 1]:
    t = l.strip()
    if t.startswith("; RMH_PHASE"):
        phase = int(t.split()[2]); continue
    if not l.startswith("\t") or t.startswith((".", ";")): continue
    regs = [int(x) for x in re.findall(r"\bv(\d+)\b", t)] + [int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", t)]
    if regs:
        mx[phase] = max(mx.get(phase, 0), max(regs))
    if "scratch_" in t:
        cnt[phase] = cnt.get(phase, 0) + 1
for ph in mx:
    print(f"phase {ph:3d}: max vgpr index {mx[ph]:4d}  scratch insts {cnt.get(ph, 0)}")
