#!/bin/bash
# round 5, job 10: transposed table rows for the column-gather contractions (p >= 4): pre = before, main = with
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4" "--order 5 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 6 --rs 4 --lo 4" "--order 5 --rs 4 --lo 4" "--order 4 --rs 4 --lo 4" "--order 3 --rs 5"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 pre main pre main 2>&1 | grep -v amdgpu.ids
done; } > $o/job10_kbench.txt 2>&1
cat $o/job10_kbench.txt
