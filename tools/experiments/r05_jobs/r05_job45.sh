#!/bin/bash
# round 5, job 45: LDS offsets of the trace step kept from the load phase (RMH_TRACE_OFFSETS) vs recomputed in phase B
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 40 --order 6 --rs 4 main troff main troff
python3 tools/kbench.py --steps 40 --order 5 --rs 4 main troff main troff
python3 tools/kbench.py --steps 40 --order 4 --rs 5 --mesh cube01_hex main troff main troff
python3 tools/kbench.py --steps 40 main troff main troff
python3 tools/kbench.py --steps 40 --lo 4 --order 6 --rs 4 main troff
python3 tools/kbench.py --steps 40 --lo 4 main troff
} 2>&1 | grep MDOFs | cut -c1-150 > $o/job45_kbench.txt
cat $o/job45_kbench.txt
