#!/bin/bash
# round 5, job 19: final kernel -- the committed profiles, the whole -m gpu suite, smoke(), the driver's bench command
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
bash tools/jobs/r05_profile.sh > $o/job19_profile.txt 2>&1; tail -3 $o/job19_profile.txt
timeout 3000 python3 -m pytest tests -m gpu -q > $o/job19_pytest.txt 2>&1; grep -E "passed|failed" $o/job19_pytest.txt | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
