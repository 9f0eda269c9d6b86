#!/bin/bash
# round 5, job 9: cycles by phase at p = 4 and p = 5 (one wavefront per workgroup)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
for cfg in "4 0 4" "4 0 5" "4 0 6"; do echo "== stamps $cfg"; python3 tools/stamps.py $cfg 2>&1 | grep -v amdgpu.ids; done > $o/job9_stamps.txt 2>&1
cat $o/job9_stamps.txt
