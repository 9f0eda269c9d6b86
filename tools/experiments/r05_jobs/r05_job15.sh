#!/bin/bash
# round 5, job 15: face rows with three quadrature points per view; stamps of the current kernels at p = 4, 5, 6
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4" "--order 5 --rs 4" "--order 4 --rs 5 --mesh cube01_hex"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 main fvg3 main fvg3 2>&1 | grep -v amdgpu.ids
done; } > $o/job15_kbench.txt 2>&1
cat $o/job15_kbench.txt
for cfg in "4 0 4" "4 0 5" "4 0 6"; do echo "== stamps $cfg"; python3 tools/stamps.py $cfg 2>&1 | grep -v amdgpu.ids; done > $o/job15_stamps.txt 2>&1
cat $o/job15_stamps.txt
