#!/bin/bash
# round 5, job 39: interior launch held until the exchange kernel is next in its queue (RMH_COMM_FIRST=1, default) vs not (0): RCCL self-loop, -rs 5 and -rs 4
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
for rs in 5 4; do for f in 0 1 0 1; do echo -n "rs $rs RMH_COMM_FIRST=$f: "; RMH_COMM_FIRST=$f python3 tools/rccl_selfloop.py --rs $rs --steps 12 2>&1 | grep -E "self_wrap 1|bit-identical" | cut -c1-60 | tr '\n' ' '; echo; done; done > $o/job39_rates.txt 2>&1
cat $o/job39_rates.txt
for f in 0 1; do
RMH_COMM_FIRST=$f rocprofv3 --kernel-trace --output-format csv -d $o/job39_$f -o t -- python3 tools/rccl_selfloop.py --rs 5 --steps 6 > $o/job39_$f.out 2> $o/job39_$f.err
python3 - $o $f <<'PY'
import csv, glob, sys
o, f = sys.argv[1], sys.argv[2]
rows = sorted(csv.DictReader(open(glob.glob(f"{o}/job39_{f}/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].replace("void rmh::", "")[:36], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])) for r in rows]
rc = [k for k in ks if "rccl" in k[0]][-18:]
it = [k for k in ks if "ho_kernel2" in k[0] and k[3] > 20_000_000][-18:]
print(f"COMM_FIRST={f}: RCCL kernel avg {sum(e - s for _, s, e, _ in rc) / len(rc) / 1e3:.1f} us, interior launch avg {sum(e - s for _, s, e, _ in it) / len(it) / 1e3:.1f} us")
i0 = max(i for i, k in enumerate(ks) if "rccl" in k[0])
t0 = ks[i0 - 6][1]
for n, s, e, g in ks[i0 - 6:i0 + 2]:
    print(f"   {n:36s} grid {g:>9d} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:8.1f} us")
PY
done > $o/job39_trace.txt 2>&1
cat $o/job39_trace.txt
find $o -name "*kernel_trace.csv" -size +2M -delete
