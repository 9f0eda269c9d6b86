#!/bin/bash
# round 5, job 6: the whole -m gpu suite, smoke(), and the driver's bench command on the current tree
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
timeout 3000 python3 -m pytest tests -m gpu -x -q > $o/job6_pytest.txt 2>&1; tail -4 $o/job6_pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/job6_bench.out 2> $o/job6_bench.err; tail -1 $o/job6_bench.out | wc -c; tail -1 $o/job6_bench.out
cp gpurun_out/bench_detail.json $o/job6_bench_detail.json
