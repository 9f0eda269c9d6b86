#!/bin/bash
# round 6: whole -m gpu suite, smoke(), default bench (the driver's round-end sequence)
set -u
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
tag=${1:-check}
timeout 3000 python3 -m pytest tests -m gpu -q -x > $o/${tag}_pytest.txt 2>&1; grep -E "passed|failed|error" $o/${tag}_pytest.txt | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err; tail -c 3800 $o/${tag}_bench.json
