#!/bin/bash
# A/B sweep of the cell kernels on one GPU box (run through gpurun): parity subset + fuzz, then one line per shape /
# option.  Usage: tools/gpu_sweep_reg.sh ["P B N [one_case.py flags ...]" ...]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "register or lds or auto_dispatch or prefix or tiny" 2>&1 | tail -3
timeout 600 env SCV_FUZZ_SEEDS=${SCV_FUZZ_SEEDS:-500} python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-40s %8.1f us %8.1f GB/s' % (r['shape'], r.get('opts'), r['median_us'], r['GBps']))"; }
if [ $# -eq 0 ]; then
  set -- "200000 4 64" "100000 4 128" "50000 4 256" "25000 4 512" "12500 4 1024" "12500 4 2048" "6250 4 4096" "25000 32 64" "2000 8 100" "12500 4 1000" "12500 4 600"
fi
for c in "$@"; do
  set -- $c
  P=$1; B=$2; N=$3; shift 3
  run --P $P --B $B --N $N "$@"
done
for n in 8 16 32 64; do run --prefix --P 200000 --N $n; run --prefix --P 200000 --N $n --tokens; done
