#!/bin/bash
# round 5, job 41: late kernel arguments read once in front of the dof rounds (RMH_HOIST_LATE) vs per round
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 40 --order 6 --rs 4 main hoist main hoist
python3 tools/kbench.py --steps 40 --order 5 --rs 4 main hoist main hoist
python3 tools/kbench.py --steps 40 --order 4 --rs 5 --mesh cube01_hex main hoist main hoist
python3 tools/kbench.py --steps 40 main hoist
python3 tools/kbench.py --steps 40 --lo 4 --order 6 --rs 4 main hoist
} 2>&1 | grep MDOFs | cut -c1-150 > $o/job41_kbench.txt
cat $o/job41_kbench.txt
