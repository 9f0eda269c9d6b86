"""Condense the rocprofv3 CSV output of tools/profile.sh into a text summary (committed under profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    files = glob.glob(os.path.join(out, pattern), recursive=True)
    return files[0] if files else None


print("== bench (un-profiled) ==")
try:
    print(open(os.path.join(out, "bench_plain.json")).read().strip())
except Exception as e:
    print("missing", e)
print("== bench (under rocprofv3 --kernel-trace --stats) ==")
try:
    print(open(os.path.join(out, "bench_traced.json")).read().strip())
except Exception as e:
    print("missing", e)

f = find("trace/**/*kernel_stats.csv")
print("\n== rocprofv3 --kernel-trace --stats: kernel_stats ==")
if f:
    for i, row in enumerate(csv.reader(open(f))):
        if i < 16:
            print(", ".join(row))
else:
    print("no kernel_stats.csv found")

# timed-region average of the dominant kernel from the per-dispatch trace (last 3*steps launches)
f = find("trace/**/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    by = defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    print("\n== per-kernel durations from kernel_trace (all launches; ms) ==")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[:90]:90s} n={len(v):5d} avg={sum(v)/len(v):9.4f} min={min(v):9.4f} max={max(v):9.4f} total={sum(v):10.3f}")
    # the timed region of bench.py: the last 3 * steps launches of each stage kernel (the warm-up launches before it
    # run colder and slower, and at p = 3 the first launch after the synchronisation that opens the region pays for
    # the power ramp: about twice the steady duration)
    try:
        steps = json.loads([l for l in open(os.path.join(out, "bench_traced.json")) if l.startswith("{")][-1])["steps"]
    except Exception:  # noqa: BLE001
        steps = None
    if steps:
        print(f"\n== stage kernels, timed region only (last {3 * steps} launches; ms) ==")
        for k, v in by.items():
            if "ho_kernel2" in k and len(v) >= 3 * steps:
                t = v[-3 * steps:]
                med = sorted(t)[len(t) // 2]
                print(f"{k[:60]:60s} n={len(t):4d} avg={sum(t)/len(t):8.4f} median={med:8.4f} first={t[0]:8.4f} max={max(t):8.4f}")
    r0 = next((r for r in rows if "ho_kernel" in r["Kernel_Name"]), None)
    if r0:
        keys = [k for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r0]
        print("ho_kernel dispatch:", {k: r0[k] for k in keys})

for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = find(f"{name}/**/*counter_collection.csv")
    print(f"\n== {name} ==")
    if not f:
        print("no counter csv")
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "ho_kernel" in k or "limit_fused" in k:
            for c, v in cs.items():
                print(f"{k[:60]:60s} {c:24s} n={len(v):4d} avg={sum(v)/len(v):.6g}")
