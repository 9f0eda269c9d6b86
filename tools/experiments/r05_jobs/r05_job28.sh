#!/bin/bash
# round 5, job 28: the layer-interleaved XCD order as built into the library (chunk from the element numbering) vs contiguous eighths (RMH_XCD_CHUNK=0)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
kb() { echo "== RMH_XCD_CHUNK=${1:-auto} : ${*:2}"; if [ -n "$1" ]; then RMH_XCD_CHUNK=$1 python3 tools/kbench.py --steps 30 "${@:2}" main; else python3 tools/kbench.py --steps 30 "${@:2}" main; fi; }
{
kb 0; kb ""; kb 658; kb 329; kb 0; kb ""
kb 0 --order 6 --rs 4; kb "" --order 6 --rs 4; kb 1152 --order 6 --rs 4
kb 0 --order 4 --rs 5 --mesh cube01_hex; kb "" --order 4 --rs 5 --mesh cube01_hex; kb 2048 --order 4 --rs 5 --mesh cube01_hex
kb 0 --order 5 --rs 4; kb "" --order 5 --rs 4
kb 0 --lo 4; kb "" --lo 4
kb 0 --lo 4 --order 6 --rs 4; kb "" --lo 4 --order 6 --rs 4
kb 0 --problem 0; kb "" --problem 0
kb 0 --order 2 --rs 5; kb "" --order 2 --rs 5
} 2>&1 | grep -v amdgpu.ids > $o/job28_kbench.txt
cat $o/job28_kbench.txt
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3 > $o/job28_pytest.txt
cat $o/job28_pytest.txt
