#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests -m gpu -x -q -k "tiny or fuzz or golden or small_n" 2>&1 | tail -6
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-30s %8.1f us %8.1f GB/s' % (r['shape'], r['opts'], r['median_us'], r['GBps']))"; }
for n in 4 8 16 32; do run --P 400000 --B 4 --N $n; run --P 400000 --B 4 --N $n --opt tiny_lane=0; done
run --P 400000 --B 4 --N 8 --tokens; run --P 400000 --B 4 --N 8 --tokens --opt tiny_lane=0
run --P 30 --B 11 --N 8 --tokens; run --P 30 --B 11 --N 8 --tokens --opt tiny_lane=0
