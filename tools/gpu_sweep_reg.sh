#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register or lds or auto_dispatch or fused_and_reduced or more_budgets" 2>&1 | tail -5
timeout 600 env SCV_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-40s %8.1f us %8.1f GB/s' % (r['shape'], r['opts'], r['median_us'], r['GBps']))"; }
for c in "200000 4 64" "100000 4 128" "50000 4 256" "25000 4 512" "12500 4 1024" "12500 4 2048" "6250 4 4096" "25000 32 64" "2000 8 100"; do
  set -- $c
  run --P $1 --B $2 --N $3
done
run --P 12500 --B 4 --N 1024 --dist 0
run --P 12500 --B 4 --N 1024 --dist 3
run --P 25000 --B 4 --N 512 --dist 0
run --P 25000 --B 4 --N 512 --dist 3
run --P 12500 --B 4 --N 1000
run --P 12500 --B 4 --N 600
run --P 6250 --B 4 --N 4096 --tokens
run --P 12500 --B 4 --N 1024 --tokens
run --P 50000 --B 4 --N 256 --tokens
