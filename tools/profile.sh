#!/bin/bash
# Profiling recipe run on the GPU box through gpurun (outputs under gpurun_out/prof_<tag>/).
#   bash tools/profile.sh <tag> [bench args...]
# Pass 1: kernel trace + stats of the bench command; passes 2-4: PMC counters, each in its own run
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; no trace domains next to --pmc).
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py "$@" --no-extras > $out/bench_plain.json 2> $out/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 bench.py "$@" --no-extras > $out/bench_traced.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- python3 bench.py "$@" --no-extras --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- python3 bench.py "$@" --no-extras --no-cpu-baseline > /dev/null 2> $out/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -o pmc -- python3 bench.py "$@" --no-extras --no-cpu-baseline > /dev/null 2> $out/pmc_sq.err
python3 tools/profile_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
# keep the merge-back small: raw per-dispatch CSVs can be large
find $out -name "*.csv" -size +8M -delete
