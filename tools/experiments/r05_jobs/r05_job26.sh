#!/bin/bash
# round 5, job 26: the load phase split by a temporary stamp (23 = until every global load is issued, incl. the wait for the neighbour indices; 0 = LDS stores + first barrier)
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
for cfg in "5 0 3" "4 0 5" "4 0 6"; do echo "== stamps $cfg"; python3 tools/stamps.py $cfg 2>&1 | grep -E "stamp 23|A loads|total per WG|B T1|traces"; done > $o/job26_stamps.txt 2>&1
cat $o/job26_stamps.txt
