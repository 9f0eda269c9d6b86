#!/bin/bash
# round 5, job 24: hierarchical mesh nodes along y as well (RMH_HIER = 7) after this round's register savings
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 3 --rs 5" "--order 6 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 5 --rs 4" "--order 3 --rs 5 --lo 4" "--order 2 --rs 5"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 main hier7 main hier7 2>&1 | grep -v amdgpu.ids
done; } > $o/job24_kbench.txt 2>&1
cat $o/job24_kbench.txt
