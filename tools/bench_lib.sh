#!/bin/bash
# bench with an alternative build of the library: bash tools/bench_lib.sh <lib.so> [bench args]
lib=$1; shift
cp remhos_amd/librmh.so /tmp/librmh_saved.so
cp $lib remhos_amd/librmh.so
python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['mass_cg_max_iters'])"
cp /tmp/librmh_saved.so remhos_amd/librmh.so
