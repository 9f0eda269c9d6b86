#!/bin/bash
# round 5, job 8: how long is the load phase of the p = 3 stage when its data is cache-resident?  (stamps at -rs 3 / 4 / 5)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
for rs in 3 4 5; do echo "== stamps rs $rs p 3"; python3 tools/stamps.py $rs 0 3 2>&1 | grep -v amdgpu.ids; done > $o/job8_stamps_rs.txt 2>&1
cat $o/job8_stamps_rs.txt
