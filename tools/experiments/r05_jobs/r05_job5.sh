#!/bin/bash
# round 5, job 5: element numbering in y-strips (rmhd_config.tile_rows): rate and HBM read bytes against the lattice order
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ echo "== --order 3 --rs 5"; python3 tools/kbench.py --order 3 --rs 5 --steps 30 main main:2 main:4 main:6 main:8 main:12 main main:4 main:8 2>&1 | grep -v amdgpu.ids
  for cfg in "--order 6 --rs 4" "--order 3 --rs 5 --lo 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 3 --rs 5 --problem 0"; do
    echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 main main:4 main:8 main main:4 main:8 2>&1 | grep -v amdgpu.ids
  done; } > $o/job5_kbench.txt 2>&1
cat $o/job5_kbench.txt
for t in 0 4 8; do
  for c in FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $o/pmc5_t${t}_$c -o pmc -- python3 tools/kbench.py --order 3 --rs 5 --steps 4 --tile $t main > /dev/null 2>&1
    python3 - $o/pmc5_t${t}_$c tile$t $c <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "ho_kernel2" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(sys.argv[2], k, "avg KiB per launch", sum(v) / len(v), "n", len(v))
PY
  done
done > $o/job5_traffic.txt 2>&1
cat $o/job5_traffic.txt
find $o -name "*.csv" -size +2M -delete
