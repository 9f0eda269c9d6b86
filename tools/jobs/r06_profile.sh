#!/bin/bash
# round 6: the committed profiles (run at the END of the kernel work: entries are keyed by the hash of the kernel sources)
set -u
export TMPDIR=/tmp
bash tools/profile.sh r06 --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/pmc_insts.sh r06 --steps 20 --warmup 5 > gpurun_out/prof_r06/pmc_insts.txt 2>&1
bash tools/profile.sh r06_lo4 --lo 4 --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/pmc_insts.sh r06_lo4 --lo 4 --steps 20 --warmup 5 > gpurun_out/prof_r06_lo4/pmc_insts.txt 2>&1
tail -n 40 gpurun_out/prof_r06/summary.txt; tail -n 30 gpurun_out/prof_r06_lo4/summary.txt
