#!/bin/bash
# round 5, job 3: CU reservation for the exchange stream on the one-rank RCCL self-loop (timeline + rates), the C++ loop with
# RMH_COMM_CUS, bench.py --gpus 2 in the one-GPU validation mode (compact line, strong leg, per-rank logs), new tests
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o/rccl
timeout 1500 python3 -m pytest tests/test_gpu_binary.py tests/test_gpu_sweeps.py -m gpu -x -q > $o/job3_pytest.txt 2>&1; tail -3 $o/job3_pytest.txt
run() { tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $o/rccl/$tag -o t -- python3 tools/rccl_selfloop.py --steps 4 "$@" > $o/rccl/$tag.log 2>&1
  echo "== $tag: $(grep self_wrap $o/rccl/$tag.log | tr '\n' ' ')"; grep bit-identical $o/rccl/$tag.log
  python3 tools/trace_timeline.py $o/rccl/$tag 8; }
{ run cus0; run cus8 --comm-cus 8; run cus16 --comm-cus 16; run cus8_rs5 --rs 5 --comm-cus 8; run cus0_rs5 --rs 5; } > $o/job3_rccl_timeline.txt 2>&1
find $o/rccl -name "*.csv" -size +2M -delete
for k in 0 8 0 8; do
  rm -f /tmp/rmh_self.id
  echo "== remhos_amd_run self-wrap RCCL loop, RMH_COMM_CUS=$k"
  RMH_COMM_CUS=$k remhos_amd/remhos_amd_run -m periodic-cube -p 10 -rs 5 -o 3 -dt -1 -tf 0.5 -ms 20 -warmup 3 -pa -self-wrap 1 -comm-file /tmp/rmh_self.id 2>&1 | grep -E "FOM wall|Final mass|error"
done > $o/job3_cpp_selfloop.txt 2>&1
RMH_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --rs 4 --steps 10 --warmup 2 > $o/job3_bench_n2_onegpu.txt 2> $o/job3_bench_n2_onegpu.err
tail -c 3000 $o/job3_bench_n2_onegpu.txt; tail -3 $o/job3_bench_n2_onegpu.err; cat gpurun_out/bench_rank0.log
cat $o/job3_rccl_timeline.txt $o/job3_cpp_selfloop.txt
# p = 6 LDS bank conflicts: row stride of U1 / M1 (S2 = D^2 + 1 is EVEN at odd D: lanes 0 and 8 of a 16-lane group share banks)
bash tools/pmc_variants.sh r05_s2pad "--order 6 --rs 4 --steps 10" main s2p2 s2p4 > $o/job3_s2pad_p6.txt 2>&1
bash tools/pmc_variants.sh r05_s2pad4 "--order 4 --rs 5 --mesh cube01_hex --steps 10" main s2p2 s2p4 > $o/job3_s2pad_p4.txt 2>&1
for cfg in "--order 6 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 2 --rs 5" "--order 6 --rs 4 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 main s2p2 s2p4 main s2p2 s2p4 2>&1 | grep -v amdgpu.ids
done > $o/job3_s2pad_kbench.txt 2>&1
cat $o/job3_s2pad_p6.txt $o/job3_s2pad_p4.txt $o/job3_s2pad_kbench.txt
