#!/bin/bash
# LDS counters of the stage kernel for several builds of the library (tools/build_variant.sh), one rocprofv3 --pmc pass each:
#   bash tools/pmc_variants.sh <tag> "<kbench args>" name1 name2 ...      ("main" = librmh.so)
# prints SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, ... per launch of every ho_kernel2 instance, and the kbench line (time).
set -u
tag=$1; shift
kargs=$1; shift
out=gpurun_out/pmcv_$tag
mkdir -p $out
export TMPDIR=/tmp
for name in "$@"; do
  python3 tools/kbench.py $kargs $name > $out/$name.kbench 2>&1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/$name -o pmc -- python3 tools/kbench.py $kargs $name > $out/$name.log 2> $out/$name.err || echo "pass $name failed"
done
python3 - "$out" "$@" <<'PY'
import csv, glob, collections, sys
out, names = sys.argv[1], sys.argv[2:]
for name in names:
    try:
        print(open(f"{out}/{name}.kbench").read().strip().split("\n")[-1])
    except Exception as e:
        print(name, "no kbench line", e)
    f = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "ho_kernel2" in k:
            print(f"  {name:12s} {k[:46]:46s} " + "  ".join(f"{c.replace('SQ_', '')}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())))
PY
find $out -name "*.csv" -size +4M -delete
