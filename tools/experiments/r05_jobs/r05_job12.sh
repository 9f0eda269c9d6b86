#!/bin/bash
# round 5, job 12: geometry pass of the column phase: a table view per group of 2 / 3 quadrature planes (main = per plane)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4" "--order 5 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 6 --rs 4 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 main g7_2 g7_3 main g7_2 g7_3 2>&1 | grep -v amdgpu.ids
done; } > $o/job12_kbench.txt 2>&1
cat $o/job12_kbench.txt
