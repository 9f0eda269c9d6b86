#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for cfg in "--order 3 --rs 5" "--order 3 --rs 5 --lo 4" "--order 2 --rs 5" "--order 1 --rs 5" "--order 4 --rs 4" "--order 5 --rs 4" "--order 6 --rs 4" "--order 6 --rs 4 --lo 4" "--order 3 --rs 4" "--order 4 --rs 5 --mesh cube01_hex"; do
  echo "== $cfg"
  python3 tools/kbench.py $cfg --steps 20 fm0 main fm0 main 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r04/job8_kbench.txt
bash tools/pmc_variants.sh p3jpad "--order 3 --rs 5 --steps 10" fm0 main > gpurun_out/r04/pmcv_p3jpad.txt 2>&1
cat gpurun_out/r04/pmcv_p3jpad.txt
