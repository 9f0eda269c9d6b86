#!/bin/bash
# round 6: the tests added this round, verbose (-s: the measured errors are printed)
set -u
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_check_violation.py tests/test_gpu_binary.py tests/test_gpu_parity.py tests/test_2d.py -m gpu -q -s -k "violation or verify_bounds or limiter_tight or configs0" > $o/newtests.txt 2>&1
grep -E "TIGHT|passed|failed|Error|error" $o/newtests.txt | tail -40
