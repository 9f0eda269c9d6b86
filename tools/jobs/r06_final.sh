#!/bin/bash
# round 6: the final tree -- whole -m gpu suite, smoke(), the driver's bench command
set -u
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
timeout 3000 python3 -m pytest tests -m gpu -q > $o/final_pytest.txt 2>&1; grep -E "passed|failed|error" $o/final_pytest.txt | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_final.out 2> $o/bench_final.err; tail -c 4200 $o/bench_final.out
