#!/bin/bash
# round 6: sensitivity of the stage kernel's time to its VALU instruction count -- builds with 80 / 160 extra v_nop per wavefront in
# the y-leg (an issue-bound phase), A/B in one process per order (tools/kbench.py).  Settles whether p = 3 (at the board's power
# limit) responds to instruction counts like p = 4 (not at the limit): VERDICT round 5, next #1.
#   build: bash tools/build_variant.sh nop80 -DRMH_NOP_PROBE=80; bash tools/build_variant.sh nop160 -DRMH_NOP_PROBE=160
bash tools/experiments/r06_jobs/ab.sh "p3 p4 p5 p6" main nop80 nop160
