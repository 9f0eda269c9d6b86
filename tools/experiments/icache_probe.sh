#!/bin/bash
# Development aid (GPU box): one PMC pass over tools/kbench.py with two builds of the library in one process -- librmh.so and
# the variant remhos_amd/librmh_nosplit.so (bash tools/build_variant.sh nosplit -DRMH_COLSPLIT=0) -- averaged per build.
# Used in round 3 to find why the first split-column kernels were slower: not the instruction cache (0.19 % vs 0.09 % misses)
# but 2.7e7 more scratch instructions per launch (PMC="SQ_INSTS_VALU ... SQ_INSTS_FLAT" bash tools/experiments/icache_probe.sh).
export TMPDIR=/tmp
rocprofv3 --pmc ${PMC:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES} --output-format csv -d gpurun_out/ic -o pmc -- python3 tools/kbench.py --order 6 --rs 4 --steps 3 main nosplit > gpurun_out/ic.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/ic/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
order=[]
for r in csv.DictReader(open(f[0])):
    if "ho_kernel2<6" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
for c, d in acc.items():
    ids=sorted(d)
    h=len(ids)//2
    a=[sum(d[i]) for i in ids[:h]]; b=[sum(d[i]) for i in ids[h:]]
    print(f"{c:22s} first lib (split) avg {sum(a)/len(a):.4g}   second lib (nosplit) avg {sum(b)/len(b):.4g}")
PY
