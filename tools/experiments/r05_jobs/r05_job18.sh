#!/bin/bash
# round 5, job 18: opaque LDS bases in the face rows (main) against without (nofopq)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4" "--order 6 --rs 4 --lo 4" "--order 5 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 5 --rs 4 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 nofopq main nofopq main 2>&1 | grep -v amdgpu.ids
done; } > $o/job18_kbench.txt 2>&1
cat $o/job18_kbench.txt
