#!/bin/bash
set -u
export TMPDIR=/tmp
python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for cfg in "--order 3 --rs 5" "--order 6 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 5 --rs 4" "--order 2 --rs 5" "--order 3 --rs 5 --lo 4" "--order 6 --rs 4 --lo 4"; do
  echo "== $cfg"
  python3 tools/kbench.py $cfg --steps 20 prev main prev main 2>&1 | grep -v amdgpu.ids
done
