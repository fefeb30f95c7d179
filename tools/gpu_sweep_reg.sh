#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register or lds or auto_dispatch or prefix" 2>&1 | tail -3
timeout 600 env SCV_FUZZ_SEEDS=500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-40s %8.1f us %8.1f GB/s' % (r['shape'], r['opts'], r['median_us'], r['GBps']))"; }
run --P 200000 --B 4 --N 64
run --P 200000 --B 4 --N 64 --opt reg_shape=1601
run --P 200000 --B 4 --N 48
run --P 200000 --B 4 --N 48 --opt reg_shape=1601
run --P 200000 --B 4 --N 64 --tokens
run --P 200000 --B 4 --N 64 --tokens --opt reg_shape=1601
run --P 200000 --B 4 --N 64 --dist 3
run --P 200000 --B 4 --N 64 --dist 3 --opt reg_shape=1601
run --P 200000 --B 4 --N 40
