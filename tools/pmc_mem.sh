#!/bin/bash
# memory-pipeline counters of the bench kernels; few counters per pass (a pass that asks for more than the
# hardware can collect aborts and rocprofv3 then hangs: every pass runs under its own timeout)
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_mem_$1; shift
mkdir -p $out
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o pmc -- python3 bench.py "$@" --no-extras --no-cpu-baseline > /dev/null 2> $out/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "ho_kernel2" in k:
            for c, v in cs.items():
                print(f"{k[:40]:40s} {c:38s} n={len(v):4d} avg={sum(v)/len(v):.6g}")
PY
find $out -name "*.csv" -size +8M -delete
