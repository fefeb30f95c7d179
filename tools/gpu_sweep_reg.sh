#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q -k "register_resident or fuzz or golden or small_n or auto_dispatch" 2>&1 | tail -3
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-40s %8.1f us %8.1f GB/s' % (r['shape'], r['opts'], r['median_us'], r['GBps']))"; }
for d in 0 1 2 3; do echo "dist $d"; run --P 200000 --B 4 --N 64 --dist $d; run --P 100000 --B 4 --N 256 --dist $d; run --P 100000 --B 4 --N 512 --dist $d; run --P 50000 --B 4 --N 1024 --dist $d; run --P 40000 --B 4 --N 2048 --dist $d; run --P 20000 --B 8 --N 4096 --dist $d; done
