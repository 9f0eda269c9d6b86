#!/bin/bash
# round 5: the N = 2, 4, 8 code paths of bench.py at the DEFAULT (weak-scaling) sizes, all blocks on one GPU (RMH_BENCH_ONE_GPU=1: device
# copies instead of RCCL; a validation of the partition / halo-first / split-launch logic at the real per-rank size, not a benchmark)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
for n in 2 4 8; do
  RMH_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus $n --steps 4 --warmup 2 > $o/onegpu_n$n.out 2> $o/onegpu_n$n.err
  echo "N=$n rc=$? bytes=$(tail -1 $o/onegpu_n$n.out | wc -c)"; tail -1 $o/onegpu_n$n.out | cut -c1-900
done
