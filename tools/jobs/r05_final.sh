#!/bin/bash
# round 5, job 23: FINAL kernel -- committed profiles, whole -m gpu suite, smoke()
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
bash tools/jobs/r05_profile.sh > $o/final_profile.txt 2>&1; tail -2 $o/final_profile.txt
timeout 3000 python3 -m pytest tests -m gpu -q > $o/final_pytest.txt 2>&1; grep -E "passed|failed" $o/final_pytest.txt | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
