#!/bin/bash
# round 5, job 38: kernel timeline of a partitioned stage of ONE rank (RCCL self-loop, -rs 5): pack, RCCL kernel, interior launch, halo shell, gaps
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
python3 tools/rccl_selfloop.py --rs 5 --steps 10 > $o/job38_plain.txt 2>&1; cat $o/job38_plain.txt | grep -v amdgpu
rocprofv3 --kernel-trace --output-format csv -d $o/job38 -o t -- python3 tools/rccl_selfloop.py --rs 5 --steps 6 > $o/job38.out 2> $o/job38.err
python3 - $o <<'PY' > $o/job38_timeline.txt 2>&1
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(f"{o}/job38/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].replace("void rmh::", "")[:44], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), r["Stream_Id"]) for r in rows]
t0 = ks[-60][1]
prev = None
for n, s, e, g, q in ks[-60:]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{n:44s} grid {g:>10d} strm {q:>3s} start {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:8.1f} us  gap-to-latest-end {gap:8.1f}")
    prev = max(prev or 0, e)
PY
cat $o/job38_timeline.txt
find $o/job38 -name "*.csv" -size +2M -delete
