#!/bin/bash
# round-5 development session 1: scv_prefix_pool v2 -- parity, then timing against the dense expansion
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== parity (prefix)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prefix" --tb=short 2>&1 | tail -15 | tee gpurun_out/r5_run1_pytest.log
echo "== fuzz 200"; SCV_FUZZ_SEEDS=200 timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x --tb=short 2>&1 | tail -5 | tee -a gpurun_out/r5_run1_pytest.log
for opts in "" "prefix_path=4 reg_shape=16" "prefix_path=4 reg_shape=32"; do
  echo "== prefix_small $opts"; timeout 300 python tools/prefix_small.py $opts 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print(d['shape'], d['tokens'], '%.1f us' % d['prefix_us'], ' dense %.1f' % d.get('dense_us', float('nan')), ' pool GB/s %.0f' % d['pool_GBps'])
"
done 2>&1 | tee gpurun_out/r5_run1_prefix.log
