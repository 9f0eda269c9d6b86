#!/bin/bash
# round 5, job 36: what the neighbour-trace loads and the face-table loads cost (timing-only builds that replace them by constants: wrong numbers, same instruction stream otherwise)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 30 main nofgc nogn noboth main
python3 tools/kbench.py --steps 30 --order 6 --rs 4 main nofgc nogn noboth main
python3 tools/kbench.py --steps 30 --order 4 --rs 5 --mesh cube01_hex main nofgc nogn noboth
} 2>&1 | grep MDOFs | cut -c1-100 > $o/job36_kbench.txt
cat $o/job36_kbench.txt
