#!/bin/bash
# round 5, job 16: old = committed kernel; exit = + one-wavefront element sums by v_readlane (p = 4, 5); main = exit + no early exit in front of the kernarg loads
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 5 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 6 --rs 4" "--order 3 --rs 5" "--order 5 --rs 4 --lo 4" "--order 4 --rs 4 --lo 4" "--order 2 --rs 5"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 old exit main old exit main 2>&1 | grep -v amdgpu.ids
done; } > $o/job16_kbench.txt 2>&1
cat $o/job16_kbench.txt
