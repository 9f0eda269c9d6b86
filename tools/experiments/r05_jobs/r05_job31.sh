#!/bin/bash
# round 5, job 31: the built-in choice (2 layers woven, whole rounds) against weave 0 / 2 and contiguous eighths; the new GPU tests
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
kb() { echo -n "${1:-default} ${*:2}: "; env $1 python3 tools/kbench.py --steps 40 "${@:2}" main 2>&1 | grep MDOFs; }
{
for rep in 1 2; do for e in RMH_XCD_CHUNK=0 RMH_XCD_WEAVE=0 RMH_X=1 RMH_XCD_WEAVE=2; do kb $e; done; done
for rep in 1 2; do for e in RMH_XCD_CHUNK=0 RMH_XCD_WEAVE=0 RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --order 6 --rs 4; done; done
for e in RMH_XCD_CHUNK=0 RMH_XCD_WEAVE=0 RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --order 4 --rs 5 --mesh cube01_hex; done
for e in RMH_XCD_CHUNK=0 RMH_XCD_WEAVE=0 RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --order 5 --rs 4; done
for e in RMH_XCD_CHUNK=0 RMH_XCD_WEAVE=0 RMH_X=1; do kb $e --lo 4; done
for e in RMH_XCD_CHUNK=0 RMH_XCD_WEAVE=0 RMH_X=1; do kb $e --lo 4 --order 6 --rs 4; done
for e in RMH_XCD_CHUNK=0 RMH_XCD_WEAVE=0 RMH_X=1; do kb $e --order 3 --rs 4; done
} > $o/job31_scan.txt 2>&1
cat $o/job31_scan.txt
python3 -m pytest tests/test_gpu_tile_order.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 > $o/job31_pytest.txt
cat $o/job31_pytest.txt
