#!/bin/bash
# prefix budgets 1, 2, 4 ... N over pools of 256 / 1024 / 4096 votes (25.6 M votes each) on the distributions D0 .. D5: scv_prefix_pool must not depend on where the hot value is
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
for N in 256 1024 4096; do for d in 0 1 2 3 4 5; do P=$((25600000/N)); python $R/tools/one_case.py --prefix --rounds 5 --P $P --N $N --dist $d 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('N', d['shape'][2], 'dist', $d, '%.1f us' % d['median_us'], '%.0f GB/s' % d['GBps'])"; done; done
