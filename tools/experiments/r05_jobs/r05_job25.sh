#!/bin/bash
# round 5, job 25: flat addressing of the batch's contiguous arrays in the load phase (main) against per-element index arithmetic (noflat)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 3 --rs 5" "--order 2 --rs 5" "--order 3 --rs 5 --lo 4" "--order 6 --rs 4" "--order 1 --rs 5"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 noflat main noflat main 2>&1 | grep -v amdgpu.ids
done; } > $o/job25_kbench.txt 2>&1
cat $o/job25_kbench.txt
