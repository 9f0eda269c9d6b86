#!/bin/bash
# round 5, job 34: speculative neighbour-trace / face-table loads at predicted addresses (HoArgs::pred_stride) vs none (RMH_PREDICT=0)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
kb() { echo -n "${1:-default} ${*:2}: "; env $1 python3 tools/kbench.py --steps 40 "${@:2}" main 2>&1 | grep MDOFs; }
{
for rep in 1 2 3; do for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e; done; done
for rep in 1 2; do for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e --order 6 --rs 4; done; done
for rep in 1 2; do for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e --order 4 --rs 5 --mesh cube01_hex; done; done
for rep in 1 2; do for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e --order 5 --rs 4; done; done
for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e --lo 4; done
for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e --lo 4 --order 6 --rs 4; done
for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e --order 2 --rs 5; done
for e in RMH_PREDICT=0 RMH_PREDICT=1; do kb $e --problem 0; done
} > $o/job34_scan.txt 2>&1
cat $o/job34_scan.txt
python3 -m pytest tests/test_gpu_tile_order.py tests/test_gpu_parity.py tests/test_gpu_selfloop.py tests/test_gpu_exchange.py -x -q -m gpu 2>&1 | grep -E "passed|failed" > $o/job34_pytest.txt
cat $o/job34_pytest.txt
