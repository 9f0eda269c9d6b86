#!/bin/bash
# round 5, job 33: every other stage launch walks the batches backwards (RMH_ALT_ORDER=1): the end of a stage's output is the freshest in the Infinity Cache
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
kb() { echo -n "${1:-default} ${*:2}: "; env $1 python3 tools/kbench.py --steps 40 "${@:2}" main 2>&1 | grep MDOFs; }
{
for rep in 1 2 3; do for e in RMH_ALT_ORDER=0 RMH_ALT_ORDER=1; do kb $e; done; done
for rep in 1 2; do for e in RMH_ALT_ORDER=0 RMH_ALT_ORDER=1; do kb $e --order 6 --rs 4; done; done
for rep in 1 2; do for e in RMH_ALT_ORDER=0 RMH_ALT_ORDER=1; do kb $e --order 4 --rs 5 --mesh cube01_hex; done; done
for rep in 1 2; do for e in RMH_ALT_ORDER=0 RMH_ALT_ORDER=1; do kb $e --order 5 --rs 4; done; done
for e in RMH_ALT_ORDER=0 RMH_ALT_ORDER=1; do kb $e --lo 4; done
for e in RMH_ALT_ORDER=0 RMH_ALT_ORDER=1; do kb $e --lo 4 --order 6 --rs 4; done
for e in RMH_ALT_ORDER=0 RMH_ALT_ORDER=1; do kb $e --order 3 --rs 4; done
} > $o/job33_scan.txt 2>&1
cat $o/job33_scan.txt
