#!/bin/bash
# round 5: the driver's bench command on the committed tree (+ the N = 2 code path in its one-GPU validation mode)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_final.out 2> $o/bench_final.err; tail -1 $o/bench_final.out | wc -c; tail -1 $o/bench_final.out
cp gpurun_out/bench_detail.json $o/bench_final_detail.json
RMH_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --rs 4 --steps 10 --warmup 2 > $o/bench_n2_onegpu.out 2> $o/bench_n2_onegpu.err; tail -1 $o/bench_n2_onegpu.out | cut -c1-600
