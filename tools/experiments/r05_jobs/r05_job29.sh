#!/bin/bash
# round 5, job 29: chunk-size scan of the layer-interleaved XCD order (RMH_XCD_CHUNK) + HBM fetch bytes per launch
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
kb() { echo -n "chunk $1 ${*:2}: "; RMH_XCD_CHUNK=$1 python3 tools/kbench.py --steps 40 "${@:2}" main 2>&1 | grep MDOFs; }
{
for rep in 1 2; do for c in 0 1316 658 439 987 1974 2633 165; do kb $c; done; done
for rep in 1 2; do for c in 0 2304 1152 768 4608 3456; do kb $c --order 6 --rs 4; done; done
for rep in 1 2; do for c in 0 4096 2048 1365 8192; do kb $c --order 4 --rs 5 --mesh cube01_hex; done; done
} > $o/job29_scan.txt 2>&1
cat $o/job29_scan.txt
for c in 0 1316 658; do
  RMH_XCD_CHUNK=$c timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/job29_f$c -o pmc -- python3 tools/kbench.py --steps 4 main > $o/job29_f$c.log 2>&1
done
python3 - $o <<'PY' > $o/job29_fetch.txt 2>&1
import csv, glob, collections, sys
o = sys.argv[1]
for name in ("0", "1316", "658"):
    f = glob.glob(f"{o}/job29_f{name}/**/*counter_collection.csv", recursive=True)
    if not f: print(name, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "ho_kernel2" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("chunk", name, k, "launches", len(v), "FETCH_SIZE mean (raw counter, KB units per the guide)", sum(v) / len(v))
PY
cat $o/job29_fetch.txt
find $o -name "*.csv" -size +4M -delete
