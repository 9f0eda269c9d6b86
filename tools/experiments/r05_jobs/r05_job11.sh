#!/bin/bash
# round 5, job 11: table views per GROUP of outputs in the pencil-type contractions: g1 = a view per output (before), g16 / main (24) / g36 = doubles in flight per view
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
{ for cfg in "--order 6 --rs 4" "--order 5 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 6 --rs 4 --lo 4" "--order 5 --rs 4 --lo 4" "--order 4 --rs 4 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 g1 g16 main g36 g1 g16 main g36 2>&1 | grep -v amdgpu.ids
done; } > $o/job11_kbench.txt 2>&1
cat $o/job11_kbench.txt
