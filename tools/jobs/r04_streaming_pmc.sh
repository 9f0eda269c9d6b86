#!/bin/bash
# HBM traffic of the streaming kernels of the granular call sequence (rmh_stream.hpp, rmh_kernels.hpp): FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes over tools/gbench.py (every kernel launched back to back), p = 3 -rs 5
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_stream
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o pmc -- python3 tools/gbench.py --order 3 --rs 5 --reps 5 main > $out/$c.log 2> $out/$c.err
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_stream/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 56623104
alg = {"elem_minmax_kernel": 8 * n + 16 * 884736, "bounds_kernel": 16 * n + 4 * 27 * 884736 + 16 * 884736, "lo_massavg_kernel": 32 * n,
       "fct_clipscale_kernel": 56 * n, "limit_fused_kernel": None}
print("kernel, launches, FETCH_SIZE KiB (avg), WRITE_SIZE KiB (avg), HBM bytes = 1024 (1.771 FETCH + WRITE) [x 2 FETCH], algorithmic bytes")
for k, cs in sorted(acc.items()):
    key = next((a for a in alg if a in k), None)
    if key is None or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    f, w = cs["FETCH_SIZE"], cs["WRITE_SIZE"]
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    a = alg[key]
    print(f"{k[:60]:60s} {len(f):3d}  {fa:12.0f} {wa:12.0f}  {1024 * (1.771 * fa + wa) / 1e9:7.3f} GB [{1024 * (2 * fa + wa) / 1e9:7.3f}]  "
          + (f"{a / 1e9:7.3f} GB" if a else "2.36 GB (RK update form) / 1.91 GB (du form): both forms are launched"))
PY
find $out -name "*.csv" -size +4M -delete
