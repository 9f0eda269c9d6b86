#!/bin/bash
# round 5, job 22: element-uniform divisions of the RD nodal weights formed once per lane (main) against once per dof round (pre)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4 --lo 4" "--order 5 --rs 4 --lo 4" "--order 4 --rs 4 --lo 4" "--order 6 --rs 4 --lo 3"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 pre main pre main 2>&1 | grep -v amdgpu.ids
done; } > $o/job22_kbench.txt 2>&1
cat $o/job22_kbench.txt
