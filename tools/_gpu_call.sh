#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
echo "== parity (prefix, sort, short, fuzz, token)"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "prefix or sort or short or random or token" 2>&1 | tail -3
fmt='import sys, json; d = json.loads(sys.stdin.read()); P, B, N = d["shape"]; print("pools %8d x %4d votes x %2d budgets  %7.1f us  %6.0f GB/s of pool bytes" % (P, N, B, d["median_us"], d["GBps"]))'
{
echo "# same box, same session: round 6's library before this change (git b09f8ac: tools/build_ab.sh) against the library of this commit;"
echo "# tools/one_case.py --prefix [--tokens] --rounds 9 --opt prefix_path=5 (budgets 1, 2, 4 ... N promised: one launch), median of 9 cold passes"
for lib in tools/ab/libscvote_base.so ""; do
  export SCV_LIB_PATH=$lib
  for tok in "" "--tokens"; do
    echo "== library: ${lib:-this commit}  $tok"
    for N in 32 64 128; do
      for P in 1920 25000 50000 100000 150000 200000 300000 400000 800000; do python tools/one_case.py --prefix $tok --rounds 9 --P $P --N $N --opt prefix_path=5 2>&1 | grep -v amdgpu.ids | python -c "$fmt"; done
    done
  done
  echo "== library: ${lib:-this commit}  dense short cells [P, 4, N] with the cell table (scv_sort_cells)"
  for N in 16 32 64; do for P in 500 4000 16000 64000 200000; do python tools/one_case.py --P $P --B 4 --N $N --rounds 9 2>&1 | grep -v amdgpu.ids | tail -1 | python -c 'import sys, json; d = json.loads(sys.stdin.read()); print("cells %8d x 4 x %3d  %7.1f us  %6.0f GB/s" % (d["shape"][0], d["shape"][2], d["median_us"], d["GBps"]))'; done; done
done
unset SCV_LIB_PATH
} > gpurun_out/prefix_fixed_cost_ab.log 2>&1
tail -5 gpurun_out/prefix_fixed_cost_ab.log
{
echo "# SCV_TOK_PASSES = token steps a wave without a sort step in the last round takes before the others get any (scv_sort_prefix2<true>): 0, 1, 2 (the library), 3, 4"
for lib in tools/ab/libscvote_tokp0.so tools/ab/libscvote_tokp1.so "" tools/ab/libscvote_tokp3.so tools/ab/libscvote_tokp4.so; do
  export SCV_LIB_PATH=$lib
  echo "== ${lib:-the library (2)}"
  for P in 100000 170000 200000 230000 300000 400000; do python tools/one_case.py --prefix --tokens --rounds 9 --P $P --N 128 --opt prefix_path=5 2>&1 | grep -v amdgpu.ids | python -c "$fmt"; done
done
unset SCV_LIB_PATH
} > gpurun_out/prefix_token_steps_ab.log 2>&1
tail -3 gpurun_out/prefix_token_steps_ab.log
echo "== wall marks"
timeout 300 python tools/sort_prefix_wall.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort_prefix_wall.log | tail -3
echo "== crossovers"
timeout 1500 python tools/crossovers.py 2>&1 | grep -v amdgpu.ids | tail -16
