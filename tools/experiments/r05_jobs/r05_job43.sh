#!/bin/bash
# round 5, job 43: geometry pass with the plane's quadrature weight read first (RMH_W_FIRST: one scalar-memory wait per plane instead of two)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 40 --order 6 --rs 4 main wfirst main wfirst
python3 tools/kbench.py --steps 40 --order 5 --rs 4 main wfirst main wfirst
python3 tools/kbench.py --steps 40 --order 4 --rs 5 --mesh cube01_hex main wfirst main wfirst
python3 tools/kbench.py --steps 40 main wfirst
python3 tools/kbench.py --steps 40 --lo 4 --order 6 --rs 4 main wfirst
} 2>&1 | grep MDOFs | cut -c1-150 > $o/job43_kbench.txt
cat $o/job43_kbench.txt
