#!/bin/bash
# N = 65 ... 128: the 128-vote shape of scv_sort_cells (one wave per SIMD: 33.8 KB of LDS per wave) against the register-resident kernels
set -u
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
for s in "200000 128" "200000 96" "200000 68" "200000 100"; do set -- $s
  for o in "sort_n_max=64" "sort_n_max=128"; do for c in "" "--no-cells"; do
    echo "N=$2 $o $c: $(timeout 120 python tools/one_case.py --P $1 --B 4 --N $2 --rounds 4 --opt $o $c 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f us %.0f GB/s' % (d['median_us'], d['GBps']))")"
  done; done
done
