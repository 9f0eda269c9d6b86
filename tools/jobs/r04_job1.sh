#!/bin/bash
# round 4, GPU call 1: lo 4 stamps + counters, LDS layout variants
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
( python3 tools/stamps.py 5 0 3 4 > gpurun_out/r04/stamps_lo4_p3.txt 2>&1 )
( python3 tools/stamps.py 4 0 6 4 > gpurun_out/r04/stamps_lo4_p6.txt 2>&1 )
( python3 tools/stamps.py 5 0 3 5 > gpurun_out/r04/stamps_lo5_p3.txt 2>&1 )
bash tools/pmc_variants.sh p3 "--order 3 --rs 5 --steps 6" main el8 el24 el2tp2 el8tp2 el24tp2 el24tp2s2 > gpurun_out/r04/pmcv_p3.txt 2>&1
bash tools/pmc_variants.sh p3lo4 "--order 3 --rs 5 --steps 6 --lo 4" main el24tp2 el8tp2 > gpurun_out/r04/pmcv_p3lo4.txt 2>&1
bash tools/pmc_variants.sh p6 "--order 6 --rs 4 --steps 6" main el24 > gpurun_out/r04/pmcv_p6.txt 2>&1
bash tools/pmc_insts.sh r04_lo4p3 --lo 4 --no-p6 --steps 6 --warmup 2 > gpurun_out/r04/pmc_lo4p3.txt 2>&1
bash tools/pmc_insts.sh r04_lo4p6 --lo 4 --order 6 --rs 4 --steps 6 --warmup 2 > gpurun_out/r04/pmc_lo4p6.txt 2>&1
tail -n 30 gpurun_out/r04/*.txt
