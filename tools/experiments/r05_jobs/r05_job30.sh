#!/bin/bash
# round 5, job 30: 2 / 4 lattice layers woven into one XCD chunk (RMH_XCD_WEAVE = log2): z-neighbours in the SAME L2
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
kb() { echo -n "weave $1 ${*:2}: "; RMH_XCD_WEAVE=$1 python3 tools/kbench.py --steps 40 "${@:2}" main 2>&1 | grep MDOFs; }
{
for rep in 1 2; do for w in 0 1 2 3; do kb $w; done; done
for rep in 1 2; do for w in 0 1 2; do kb $w --order 6 --rs 4; done; done
for rep in 1 2; do for w in 0 1 2; do kb $w --order 4 --rs 5 --mesh cube01_hex; done; done
for w in 0 1 2; do kb $w --lo 4; done
} > $o/job30_scan.txt 2>&1
cat $o/job30_scan.txt
for w in 1 2; do
  RMH_XCD_WEAVE=$w timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/job30_f$w -o pmc -- python3 tools/kbench.py --steps 4 main > $o/job30_f$w.log 2>&1
done
python3 - $o <<'PY' > $o/job30_fetch.txt 2>&1
import csv, glob, collections, sys
o = sys.argv[1]
for name in ("1", "2"):
    f = glob.glob(f"{o}/job30_f{name}/**/*counter_collection.csv", recursive=True)
    if not f: print(name, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "ho_kernel2" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("weave", name, k, "launches", len(v), "FETCH_SIZE mean", sum(v) / len(v))
PY
cat $o/job30_fetch.txt
find $o -name "*.csv" -size +1M -delete
