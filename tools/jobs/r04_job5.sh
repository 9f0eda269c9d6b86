#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python3 -m pytest tests -x -q -m gpu > gpurun_out/r04/job5_tests.txt 2>&1
tail -3 gpurun_out/r04/job5_tests.txt
for cfg in "--order 3 --rs 5" "--order 3 --rs 5 --lo 4" "--order 3 --rs 5 --lo 3" "--order 2 --rs 5" "--order 2 --rs 5 --lo 4" "--order 3 --rs 4" "--order 3 --rs 5 --exact"; do
  echo "== $cfg"
  python3 tools/kbench.py $cfg --steps 20 base nowd main 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/job5_kbench.txt 2>&1
cat gpurun_out/r04/job5_kbench.txt
