#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-30s %8.1f us %8.1f GB/s' % (r['shape'], r['opts'], r['median_us'], r['GBps']))"; }
for sk in 0 1 0 1; do run --P 128 --B 8 --N 1048576 --tokens --opt tok_skew=$sk; done
for sk in 0 1; do run --P 2000 --B 4 --N 65536 --tokens --opt tok_skew=$sk; run --P 8000 --B 4 --N 8192 --tokens --opt tok_skew=$sk; done
timeout 900 python -m pytest tests -m gpu -x -q -k "host_mode_bit_exact or every_kernel_variant or fuzz or negative_tokens or not_16_byte" 2>&1 | tail -3
