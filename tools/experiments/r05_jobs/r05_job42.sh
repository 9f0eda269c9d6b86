#!/bin/bash
# round 5, job 42: late kernel arguments read once, pinned in front of the exec-masked blocks (main) vs per block / per round (prev)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 40 prev main prev main
python3 tools/kbench.py --steps 40 --order 6 --rs 4 prev main prev main
python3 tools/kbench.py --steps 40 --order 5 --rs 4 prev main prev main
python3 tools/kbench.py --steps 40 --order 4 --rs 5 --mesh cube01_hex prev main prev main
python3 tools/kbench.py --steps 40 --lo 4 prev main
python3 tools/kbench.py --steps 40 --lo 4 --order 6 --rs 4 prev main
python3 tools/kbench.py --steps 40 --order 2 --rs 5 prev main
} 2>&1 | grep MDOFs | cut -c1-150 > $o/job42_kbench.txt
cat $o/job42_kbench.txt
