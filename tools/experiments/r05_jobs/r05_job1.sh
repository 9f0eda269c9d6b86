#!/bin/bash
# round 5, job 1: where the stage kernel's cycles go now (stamps), and the split form of the lo 4 stage against the one-kernel form
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
for cfg in "5 0 3" "4 0 6" "5 0 3 4" "4 0 6 4"; do echo "== stamps $cfg"; python3 tools/stamps.py $cfg 2>&1 | grep -v amdgpu.ids; done > $o/job1_stamps.txt 2>&1
for cfg in "--order 6 --rs 4 --lo 4" "--order 5 --rs 4 --lo 4" "--order 4 --rs 4 --lo 4" "--order 3 --rs 5 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 20 main main@split main main@split 2>&1 | grep -v amdgpu.ids
done > $o/job1_lo4_split.txt 2>&1
cat $o/job1_stamps.txt $o/job1_lo4_split.txt
