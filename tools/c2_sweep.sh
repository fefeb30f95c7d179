#!/bin/bash
# Geometry sweep for the few-huge-cell shapes (C2 30x8x2^17, 30x2^20, 1x2^24): default vs 8 loads in flight per lane vs two
# 512-thread workgroups per CU on half-cells.  Result (dead ends): the defaults are the fastest.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-40s %8.1f us %8.1f GB/s' % (r['shape'], r.get('opts'), r['median_us'], r['GBps']))"; }
for shape in "30 8 131072" "30 1 1048576" "1 1 16777216"; do
  set -- $shape
  echo "== default"; run --P $1 --B $2 --N $3 --rounds 8
  echo "== U8"; SCV_COPIES=16 SCV_THREADS=1024 SCV_WG_PER_CU=1 SCV_UNROLL=8 run --P $1 --B $2 --N $3 --rounds 8
  echo "== U8 path2 auto segs"; SCV_COPIES=16 SCV_THREADS=1024 SCV_WG_PER_CU=1 SCV_UNROLL=8 run --P $1 --B $2 --N $3 --rounds 8 --opt path=2
  echo "== T512x2 segs2 U4"; SCV_COPIES=16 SCV_THREADS=512 SCV_WG_PER_CU=2 SCV_UNROLL=4 run --P $1 --B $2 --N $3 --rounds 8 --opt path=2 --opt segs=2
  echo "== T512x2 segs2 U8"; SCV_COPIES=16 SCV_THREADS=512 SCV_WG_PER_CU=2 SCV_UNROLL=8 run --P $1 --B $2 --N $3 --rounds 8 --opt path=2 --opt segs=2
done
