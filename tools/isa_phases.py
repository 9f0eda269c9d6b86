"""Development aid: static instruction mix of a stage kernel per phase.  Compiles rmh_api.hip to assembly with a comment
at every phase boundary (-DRMH_PHASE_MARKS: the RMH_STAMP points of rmh_ho2.hpp) and classifies the instructions between
them.  The kernel is straight-line code but for the PCG loop and a few 1-2 trip task loops, so the static count per
wavefront is close to the executed one (cross-check: SQ_INSTS_VALU / SQ_WAVES from tools/pmc_insts.sh).

    python tools/isa_phases.py [order] [mode]        (needs hipcc; no GPU)
"""
import collections
import os
import subprocess
import sys

order = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/rmh_phases.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans", "-DRMH_PHASE_MARKS", *sys.argv[3:], "-S",
                       "--cuda-device-only", os.path.join(root, "remhos_amd", "csrc", "rmh_api.hip"), "-o", out],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
kern = f"_ZN3rmh10ho_kernel2ILi{order}ELi{mode}EEEvNS_6HoArgsE"
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
NAMES = {-1: "A loads", 0: "B pencils", 1: "B face rows", 2: "C column", 3: "F y-leg", 4: "G dof x-leg + faces", 5: "I PCG prelude",
         9: "I sA write", 10: "I x-leg", 11: "I column", 12: "I y-back", 13: "I x-back", 8: "I dot 1 + update", 14: "I dot 2", 15: "I tail",
         6: "I exit + completion", 16: "J back-transform", 21: "K mass sums", 22: "K -", 17: "K bounds + clip", 18: "K pos/neg sums",
         19: "K scale + stores", 20: "K extrema", 7: "end",
         23: "B traces -> LDS (lo 4: after the subcell pass)", 24: "B traces -> LDS", 25: "B face rows", 26: "B lumped face fluxes (lo 4)", 27: "G RD: z conversion",
         28: "G RD: element sums", 29: "G RD: extrema + chunk sums", 30: "G RD: gather + weights"}


def cls(op):
    if op.startswith("v_") and "f64" in op:
        return "fp64"
    if op.startswith("v_mov") or "dpp" in op or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")):
        return "mov/lane"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"):
        return "select"
    if op.startswith("v_"):
        return "int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    return "scalar"


phase = -1
acc = collections.OrderedDict()
for l in lines[start:end + 1]:
    t = l.strip()
    if t.startswith("; RMH_PHASE"):
        phase = int(t.split()[2])
        continue
    if not l.startswith("\t") or t.startswith((".", ";")):
        continue
    acc.setdefault(phase, collections.Counter())[cls(t.split()[0])] += 1
cols = ["fp64", "int", "mov/lane", "select", "lds", "vmem", "scalar", "wait", "barrier"]
print(f"ho_kernel2<{order}, {mode}>: static instructions per wavefront between the phase marks")
print(f"{'phase':24s}" + "".join(f"{c:>9s}" for c in cols) + "    VALU  non-fp64 share")
tot = collections.Counter()
for ph, c in acc.items():
    valu = c["fp64"] + c["int"] + c["mov/lane"] + c["select"]
    print(f"{NAMES.get(ph, str(ph)):24s}" + "".join(f"{c[k]:9d}" for k in cols) + f"  {valu:6d}  {100.0 * (valu - c['fp64']) / max(valu, 1):5.1f}%")
    tot.update(c)
valu = tot["fp64"] + tot["int"] + tot["mov/lane"] + tot["select"]
print(f"{'total':24s}" + "".join(f"{tot[k]:9d}" for k in cols) + f"  {valu:6d}  {100.0 * (valu - tot['fp64']) / valu:5.1f}%")
