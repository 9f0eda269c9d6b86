#!/bin/bash
# round 5, job 7: lo 4 stage at p = 6 with its work region as dynamic LDS, so that the launch bound of 3 wavefronts per SIMD is honoured
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4 --lo 4" "--order 6 --rs 4 --lo 4 --mesh cube01_hex" "--order 6 --rs 4" "--order 3 --rs 5" "--order 3 --rs 5 --lo 4" "--order 4 --rs 5 --mesh cube01_hex"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 main dyn63 main dyn63 2>&1 | grep -v amdgpu.ids
done; } > $o/job7_kbench.txt 2>&1
cat $o/job7_kbench.txt
timeout 900 python3 -m pytest tests/test_gpu_sweeps.py tests/test_gpu_tile_order.py tests/test_gpu_binary.py -m gpu -x -q 2>&1 | tail -3
# how long is the load phase when the data is cache-resident?  (stamps at -rs 3 / 4 / 5, -pa rule, p = 3)
for rs in 3 4 5; do echo "== stamps rs $rs p 3"; python3 tools/stamps.py $rs 0 3 2>&1 | grep -E "A loads|traces|total per WG"; done > $o/job7_stamps_rs.txt 2>&1
cat $o/job7_stamps_rs.txt
