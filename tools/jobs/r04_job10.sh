#!/bin/bash
set -u
export TMPDIR=/tmp
out=gpurun_out/r04/rccl
mkdir -p $out
run() { # tag, env..., args
  tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $out/$tag -o t -- python3 tools/rccl_selfloop.py --steps 4 "$@" > $out/$tag.log 2>&1
  echo "== $tag: $(grep self_wrap $out/$tag.log | tr '\n' ' ')"
  python3 tools/trace_timeline.py $out/$tag 8
}
run default
run nooverlap --no-overlap
export NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
run ch1
run ch1_nooverlap --no-overlap
unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS
run rs5 --rs 5
find $out -name "*.csv" -size +2M -delete
