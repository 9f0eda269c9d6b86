#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for cfg in "--order 3 --rs 5" "--order 6 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 5 --rs 4" "--order 2 --rs 5" "--order 1 --rs 5" "--order 3 --rs 5 --lo 4" "--order 6 --rs 4 --lo 4" "--order 3 --rs 5 --lo 3"; do
  echo "== $cfg"
  python3 tools/kbench.py $cfg --steps 30 fm0 main fm0 main 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r04/job13_kbench.txt
