#!/bin/bash
# round 6: A/B of library builds in one process per configuration (tools/kbench.py); usage: ab.sh "<cases>" name1 name2 ...
#   cases: any of p3 p3lo4 p2 p4 p5 p6 p6lo4 p4lo4 p5lo4 (space separated)
set -u
export TMPDIR=/tmp
cases=$1; shift
for c in $cases; do
  case $c in
    p3)    args="--order 3 --rs 5";;
    p3lo4) args="--order 3 --rs 5 --lo 4";;
    p2)    args="--order 2 --rs 5";;
    p4)    args="--order 4 --rs 5 --mesh cube01_hex";;
    p5)    args="--order 5 --rs 4";;
    p6)    args="--order 6 --rs 4";;
    p6lo4) args="--order 6 --rs 4 --lo 4";;
    p4lo4) args="--order 4 --rs 5 --mesh cube01_hex --lo 4";;
    p5lo4) args="--order 5 --rs 4 --lo 4";;
  esac
  echo "== $c"
  python3 tools/kbench.py $args --steps ${STEPS:-40} "$@" "$@" 2>&1 | grep -v "amdgpu.ids"
done
