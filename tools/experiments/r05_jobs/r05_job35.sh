#!/bin/bash
# round 5, job 35: speculative neighbour loads against the previous kernel (librmh_prev.so) on one box; phase stamps of the new load phase
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{
python3 tools/kbench.py --steps 40 prev main prev main
python3 tools/kbench.py --steps 40 --order 6 --rs 4 prev main prev main
python3 tools/kbench.py --steps 40 --order 4 --rs 5 --mesh cube01_hex prev main prev main
python3 tools/kbench.py --steps 40 --order 5 --rs 4 prev main
python3 tools/kbench.py --steps 40 --lo 4 prev main
python3 tools/kbench.py --steps 40 --lo 4 --order 6 --rs 4 prev main
} 2>&1 | grep MDOFs > $o/job35_kbench.txt
cat $o/job35_kbench.txt
for cfg in "5 0 3" "4 0 6"; do echo "== stamps $cfg"; python3 tools/stamps.py $cfg 2>&1 | grep -E "A loads|total per WG|B T1|traces|pencils"; done > $o/job35_stamps.txt 2>&1
cat $o/job35_stamps.txt
