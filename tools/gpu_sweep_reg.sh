#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefix or golden" 2>&1 | tail -3
timeout 600 env SCV_FUZZ_SEEDS=500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-40s %8.1f us %8.1f GB/s' % (r['shape'], r['opts'], r['median_us'], r['GBps']))"; }
for n in 8 16 32 64; do
run --prefix --P 200000 --N $n
run --prefix --P 200000 --N $n --tokens
run --prefix --P 200000 --N $n --tokens --opt prefix_stage=0
done
