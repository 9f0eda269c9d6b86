#!/bin/bash
# round 5, job 2: the new collected tests (binary, sweeps, soak) + A/B of the limiter-phase diet (base = round 4's kernel)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_binary.py tests/test_gpu_sweeps.py -m gpu -x -q > $o/job2_pytest.txt 2>&1; tail -5 $o/job2_pytest.txt
for cfg in "--order 3 --rs 5" "--order 6 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 3 --rs 5 --lo 4" "--order 2 --rs 5" "--order 5 --rs 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 base main base main 2>&1 | grep -v amdgpu.ids
done > $o/job2_kbench.txt 2>&1
cat $o/job2_kbench.txt
bash tools/pmc_insts.sh r05_diet1 --steps 5 --warmup 2 > $o/job2_pmc.txt 2>&1; grep "ho_kernel2<3" $o/job2_pmc.txt | head -20
