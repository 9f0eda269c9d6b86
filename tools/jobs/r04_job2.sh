#!/bin/bash
# round 4, GPU call 2: LDS bank conflicts per phase (builds that end at a phase mark)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash tools/pmc_variants.sh p3stop "--order 3 --rs 5 --steps 3" stop0 stop1 stop25 stop26 stop31 stop3 stop4 stop27 stop6 stop16 stop17 main > gpurun_out/r04/pmcv_p3stop.txt 2>&1
bash tools/pmc_variants.sh p6stop "--order 6 --rs 4 --steps 3" stop0 stop1 stop25 stop26 stop31 stop3 stop4 stop27 stop6 stop16 stop17 main > gpurun_out/r04/pmcv_p6stop.txt 2>&1
cat gpurun_out/r04/pmcv_p3stop.txt gpurun_out/r04/pmcv_p6stop.txt
