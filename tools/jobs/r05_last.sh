#!/bin/bash
# round 5: last check of the committed tree -- whole -m gpu suite, smoke(), the driver's bench command
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
timeout 3000 python3 -m pytest tests -m gpu -q > $o/last_pytest.txt 2>&1; grep -E "passed|failed" $o/last_pytest.txt | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_last.out 2> $o/bench_last.err; tail -1 $o/bench_last.out | wc -c; tail -1 $o/bench_last.out | cut -c1-700
