#!/bin/bash
# round 5, job 27: XCD map with lattice layers dealt round-robin (RMH_XCD_CHUNK = batches per layer) vs contiguous eighths
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 30 main xc1317 xc658 xc2634 main xc1317
python3 tools/kbench.py --order 6 --rs 4 --steps 30 main xc2304 main xc2304
python3 tools/kbench.py --order 4 --rs 5 --mesh cube01_hex --steps 30 main xc1024 main xc1024
} > $o/job27_kbench.txt 2>&1
cat $o/job27_kbench.txt
for name in main xc1317; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $o/job27_$name -o pmc -- python3 tools/kbench.py --steps 5 $name > $o/job27_$name.log 2>&1
done
python3 - $o <<'PY'
import csv, glob, collections, sys
o = sys.argv[1]
for name in ("main", "xc1317"):
    f = glob.glob(f"{o}/job27_{name}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "ho_kernel2" in k:
            print(name, k[:40], {c: sum(v) / len(v) for c, v in cs.items()})
PY
find $o -name "*.csv" -size +4M -delete
