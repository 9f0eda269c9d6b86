#!/bin/bash
# round 5, job 37: kernel timeline of a partitioned stage (2x2x2 blocks of 96^3 on one GPU): where the 6 % over one block go
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
RMH_BENCH_ONE_GPU=1 rocprofv3 --kernel-trace --stats --output-format csv -d $o/job37 -o t -- python3 bench.py --gpus 8 --steps 3 --warmup 1 > $o/job37.out 2> $o/job37.err
python3 - $o <<'PY' > $o/job37_timeline.txt 2>&1
import csv, glob, sys, collections
o = sys.argv[1]
f = glob.glob(f"{o}/job37/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
# last 400 kernels: print name, grid, duration, gap to previous end
tail = ks[-150:]
prev = None
for n, s, e, g, q in tail:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{n:60s} grid {g:>10s} q {q:>3s} dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us")
    prev = max(prev or 0, e)
acc = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, g, q in ks[len(ks) // 2:]:
    acc[(n, g)][0] += 1; acc[(n, g)][1] += (e - s) / 1e3
print()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{k[0]:60s} grid {k[1]:>10s} n {v[0]:5d} total {v[1]:10.1f} us avg {v[1] / v[0]:9.1f}")
PY
tail -60 $o/job37_timeline.txt
find $o/job37 -name "*.csv" -size +2M -delete
