#!/bin/bash
# round 5, job 44: the batch map's divisions by multiplication (rounds and 2^32 / chunk from the host) vs runtime divisions
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 40 prev main prev main
python3 tools/kbench.py --steps 40 --order 6 --rs 4 prev main prev main
python3 tools/kbench.py --steps 40 --order 5 --rs 4 prev main prev main
python3 tools/kbench.py --steps 40 --order 4 --rs 5 --mesh cube01_hex prev main prev main
} 2>&1 | grep MDOFs | cut -c1-150 > $o/job44_kbench.txt
cat $o/job44_kbench.txt
for cfg in "5 0 3" "4 0 6" "4 0 5"; do echo "== stamps $cfg"; python3 tools/stamps.py $cfg 2>&1 | grep -E "A loads|total per WG"; done
python3 -m pytest tests/test_gpu_tile_order.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
