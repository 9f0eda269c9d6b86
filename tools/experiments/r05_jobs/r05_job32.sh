#!/bin/bash
# round 5, job 32: four woven layers (RMH_XCD_WEAVE=2) against the default two, on the multi-element-batch workloads (p <= 3)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
kb() { echo -n "${1:-default} ${*:2}: "; env $1 python3 tools/kbench.py --steps 40 "${@:2}" main 2>&1 | grep MDOFs; }
{
for rep in 1 2 3; do for e in RMH_X=1 RMH_XCD_WEAVE=2 RMH_XCD_WEAVE=3; do kb $e; done; done
for rep in 1 2; do for e in RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --lo 4; done; done
for rep in 1 2; do for e in RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --problem 0; done; done
for rep in 1 2; do for e in RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --order 2 --rs 5; done; done
for rep in 1 2; do for e in RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --order 1 --rs 6; done; done
for rep in 1 2; do for e in RMH_X=1 RMH_XCD_WEAVE=2; do kb $e --order 3 --rs 5 --mesh cube01_hex; done; done
} > $o/job32_scan.txt 2>&1
cat $o/job32_scan.txt
