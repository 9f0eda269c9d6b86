#!/bin/bash
# round 4, GPU call 3: new lo 4 LDS layout (7 elements per workgroup at p = 3): parity + A/B
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r04/job3_tests.txt 2>&1
tail -5 gpurun_out/r04/job3_tests.txt
for cfg in "--order 3 --rs 5 --lo 4" "--order 6 --rs 4 --lo 4" "--order 4 --rs 4 --lo 4" "--order 5 --rs 4 --lo 4" "--order 2 --rs 5 --lo 4" "--order 3 --rs 5 --lo 3" "--order 3 --rs 5"; do
  echo "== $cfg"
  python3 tools/kbench.py $cfg --steps 10 base main nb6 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/job3_kbench.txt 2>&1
cat gpurun_out/r04/job3_kbench.txt
