#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash tools/pmc_variants.sh p3fm "--order 3 --rs 5 --steps 10" fm0 main fmel8 fmel24 fmel8tp2 > gpurun_out/r04/pmcv_p3fm.txt 2>&1
bash tools/pmc_variants.sh p3lo4fm "--order 3 --rs 5 --steps 10 --lo 4" fm0 main fmel8 fmel24 > gpurun_out/r04/pmcv_p3lo4fm.txt 2>&1
cat gpurun_out/r04/pmcv_p3fm.txt gpurun_out/r04/pmcv_p3lo4fm.txt
python3 tools/kbench.py --order 3 --rs 5 --steps 20 fm0 main fm0 main fmel8 fmel24 2>&1 | grep -v amdgpu.ids
python3 tools/kbench.py --order 2 --rs 5 --steps 20 fm0 main fmel8 2>&1 | grep -v amdgpu.ids
