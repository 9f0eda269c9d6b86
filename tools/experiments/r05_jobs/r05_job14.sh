#!/bin/bash
# round 5, job 14: face rows: table rows of two quadrature points per view (fvg2) against one (main)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4" "--order 5 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 5 --rs 4 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 main fvg2 main fvg2 2>&1 | grep -v amdgpu.ids
done; } > $o/job14_kbench.txt 2>&1
cat $o/job14_kbench.txt
