#!/bin/bash
set -u
export TMPDIR=/tmp
for cfg in "--order 3 --rs 5" "--order 3 --rs 5 --lo 4" "--order 2 --rs 5" "--order 1 --rs 5"; do
  echo "== $cfg"
  python3 tools/kbench.py $cfg --steps 20 prev main prev main 2>&1 | grep -v amdgpu.ids
done
bash tools/pmc_variants.sh p3tpe "--order 3 --rs 5 --steps 10" prev main 2>&1
