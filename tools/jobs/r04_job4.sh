#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
( time python3 -m pytest tests -x -q -m gpu ) > gpurun_out/r04/job4_tests.txt 2>&1
tail -5 gpurun_out/r04/job4_tests.txt
for cfg in "--order 2 --rs 5 --lo 4" "--order 2 --rs 5"; do
  echo "== $cfg"
  python3 tools/kbench.py $cfg --steps 10 base main nb2_9 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/job4_kbench.txt 2>&1
cat gpurun_out/r04/job4_kbench.txt
