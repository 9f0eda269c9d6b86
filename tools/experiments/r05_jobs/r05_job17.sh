#!/bin/bash
# round 5, job 17: whole -m gpu suite on the final kernel, then the committed profiles
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
timeout 3000 python3 -m pytest tests -m gpu -q > $o/job17_pytest.txt 2>&1; tail -4 $o/job17_pytest.txt
bash tools/jobs/r05_profile.sh > $o/job17_profile.txt 2>&1; tail -3 $o/job17_profile.txt
