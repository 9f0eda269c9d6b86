#!/bin/bash
# round 5, job 4: face speed table with one block per FACE (main) against one per element side (diet1): rates, bit-identity,
# HBM read bytes (FETCH_SIZE), and the parity tests on the new table
set -u
export TMPDIR=/tmp
o=gpurun_out/r05; mkdir -p $o
# diet1 = per-side table; noopq = per-face table + odd S2; main = noopq + opaque LDS bases (p >= 4) + bound_ctrl DPP; w63 = main with 3 wavefronts per SIMD asked for the lo 4 stage at p = 6
for cfg in "--order 3 --rs 5" "--order 3 --rs 5 --problem 0" "--order 3 --rs 5 --lo 4" "--order 2 --rs 5"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 diet1 main diet1 main 2>&1 | grep -v amdgpu.ids
done > $o/job4_kbench.txt 2>&1
for cfg in "--order 6 --rs 4" "--order 4 --rs 5 --mesh cube01_hex" "--order 5 --rs 4" "--order 4 --rs 4 --lo 4" "--order 5 --rs 4 --lo 4"; do
  echo "== $cfg"; python3 tools/kbench.py $cfg --steps 30 diet1 noopq main diet1 noopq main 2>&1 | grep -v amdgpu.ids
done >> $o/job4_kbench.txt 2>&1
echo "== --order 6 --rs 4 --lo 4" >> $o/job4_kbench.txt
python3 tools/kbench.py --order 6 --rs 4 --lo 4 --steps 30 diet1 noopq main w63 diet1 noopq main w63 2>&1 | grep -v amdgpu.ids >> $o/job4_kbench.txt
cat $o/job4_kbench.txt
for name in diet1 main; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $o/pmc4_${name}_$c -o pmc -- python3 tools/kbench.py --order 3 --rs 5 --steps 4 $name > /dev/null 2>&1
    python3 - $o/pmc4_${name}_$c $name $c <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "ho_kernel2" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(sys.argv[2], k, "avg KiB per launch", sum(v) / len(v), "n", len(v))
PY
  done
done > $o/job4_traffic.txt 2>&1
cat $o/job4_traffic.txt
find $o -name "*.csv" -size +2M -delete
timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_kat.py tests/test_gpu_sweeps.py tests/test_gpu_binary.py tests/test_gpu_selfloop.py tests/test_gpu_exchange.py tests/test_gpu_multirank.py tests/test_gpu_invariants.py -m gpu -x -q > $o/job4_pytest.txt 2>&1; tail -5 $o/job4_pytest.txt
