#!/bin/bash
# instruction-mix counters of the bench kernels (two PMC passes of 8 SQ counters each)
set -u
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $out/a -o pmc -- python3 bench.py "$@" --no-extras --no-cpu-baseline > /dev/null 2> $out/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/b -o pmc -- python3 bench.py "$@" --no-extras --no-cpu-baseline > /dev/null 2> $out/b.err
python3 - <<PY
import csv, glob, collections
for d in ("a","b"):
    f = glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print("no csv", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "ho_kernel" in k or "limit_fused" in k:
            for c, v in cs.items():
                print(f"{k[:50]:50s} {c:26s} n={len(v):4d} avg={sum(v)/len(v):.6g}")
PY
find $out -name "*.csv" -size +8M -delete
