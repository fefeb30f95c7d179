#!/bin/bash
cd /root/repo
bash tools/gpu_round.sh > gpurun_out/gpu_round_r06.log 2>&1
tail -3 gpurun_out/gpu_round_r06.log
fmt='import sys, json; d = json.loads(sys.stdin.read()); P, B, N = d["shape"]; print("pools %8d x %4d votes x %2d budgets  %7.1f us  %6.0f GB/s of pool bytes" % (P, N, B, d["median_us"], d["GBps"]))'
{
echo "# same box, same session: round 6's library before the prefix-launch work (git b09f8ac: tools/build_ab.sh) against the final library;"
echo "# tools/one_case.py --prefix [--tokens] --rounds 9 --opt prefix_path=5 (budgets 1, 2, 4 ... N promised: one launch), median of 9 cold passes"
for lib in tools/ab/libscvote_base.so ""; do
  export SCV_LIB_PATH=$lib
  for tok in "" "--tokens"; do
    echo "== library: ${lib:-final}  $tok"
    for N in 32 64 128; do
      for P in 1920 25000 50000 100000 150000 200000 300000 400000 800000; do python tools/one_case.py --prefix $tok --rounds 9 --P $P --N $N --opt prefix_path=5 2>&1 | grep -v amdgpu.ids | python -c "$fmt"; done
    done
  done
  echo "== library: ${lib:-final}  dense short cells [P, 4, N] with the cell table (scv_sort_cells)"
  for N in 16 32 64; do for P in 500 4000 16000 64000 200000; do python tools/one_case.py --P $P --B 4 --N $N --rounds 9 2>&1 | grep -v amdgpu.ids | tail -1 | python -c 'import sys, json; d = json.loads(sys.stdin.read()); print("cells %8d x 4 x %3d  %7.1f us  %6.0f GB/s" % (d["shape"][0], d["shape"][2], d["median_us"], d["GBps"]))'; done; done
done
unset SCV_LIB_PATH
} > gpurun_out/prefix_fixed_cost_ab.log 2>&1
tail -3 gpurun_out/prefix_fixed_cost_ab.log
{
echo "== extended fuzz on the FINAL library of round 6: seeds 800 .. 20799 of test_random_configuration_is_bit_exact, 800 .. 10799 of the prefix fuzz"
SCV_FUZZ_FIRST=800 SCV_FUZZ_SEEDS=20000 SCV_FUZZ_PREFIX_SEEDS=10000 SCV_FUZZ_CELL_SEEDS=1 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "not short_cells" 2>&1 | tail -2
echo "== ... 4000 further seeds of test_random_short_cells_in_device_memory_both_record_forms"
SCV_FUZZ_FIRST=800 SCV_FUZZ_SEEDS=1 SCV_FUZZ_PREFIX_SEEDS=1 SCV_FUZZ_CELL_SEEDS=4000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "short_cells" 2>&1 | tail -2
} > gpurun_out/fuzz_extended.log 2>&1
cat gpurun_out/fuzz_extended.log
