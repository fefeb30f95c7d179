#!/bin/bash
# scv_sort_prefix (budgets 1, 2, 4 ... 64 over pools of 64 votes, promised: one launch) and scv_prefix_pool (128 votes) against the number of pools:
# the fixed part of a launch and its steady-state rate
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
fmt='import sys, json; d = json.loads(sys.stdin.read()); P, B, N = d["shape"]; print("pools %8d x %4d votes x %2d budgets  %7.1f us  %6.0f GB/s of pool bytes" % (P, N, B, d["median_us"], d["GBps"]))'
for P in 25000 50000 100000 200000 400000 800000 1600000; do python $R/tools/one_case.py --prefix --rounds 5 --P $P --N 64 --opt prefix_path=5 2>&1 | grep -v amdgpu.ids | python -c "$fmt"; done
for P in 25000 50000 100000 200000 400000 800000 1600000; do python $R/tools/one_case.py --prefix --rounds 5 --P $P --N 32 --opt prefix_path=5 2>&1 | grep -v amdgpu.ids | python -c "$fmt"; done
for P in 50000 200000 800000; do python $R/tools/one_case.py --prefix --rounds 5 --P $P --N 128 2>&1 | grep -v amdgpu.ids | python -c "$fmt"; done
for P in 25000 100000 400000; do python $R/tools/one_case.py --prefix --rounds 5 --P $P --N 256 2>&1 | grep -v amdgpu.ids | python -c "$fmt"; done
