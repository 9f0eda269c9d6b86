#!/bin/bash
# round 5, job 20: element scalars of one-element workgroups formed once per lane (main) instead of once per dof round (pre)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 5 --rs 4" "--order 6 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 6 --rs 4 --lo 4" "--order 5 --rs 4 --lo 4" "--order 4 --rs 4 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 pre main pre main 2>&1 | grep -v amdgpu.ids
done; } > $o/job20_kbench.txt 2>&1
cat $o/job20_kbench.txt
