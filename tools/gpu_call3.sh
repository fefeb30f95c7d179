#!/bin/bash
# round 4, third GPU session: C2 geometry (probe + kernel sweep), tokens A/B against the round-3 library on the same box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== token tests"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 --tb=short -k "token or kernel_variant or overwrite or host_mode or unaligned or not_16" 2>&1 | tail -5
echo "== tokens A/B: round-3 library"; SCV_LIB_PATH=$R/tools/ab/libscvote_r03.so timeout 600 python tools/regimes.py --only="tokens stream" --only="headline" --only="C2" --only="N=4096 + tokens" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tokens_r03.log
echo "== tokens A/B: this build"; timeout 600 python tools/regimes.py --only="tokens stream" --only="headline" --only="C2" --only="N=4096 + tokens" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tokens_r04.log
echo "== tokens A/B again: round-3 library"; SCV_LIB_PATH=$R/tools/ab/libscvote_r03.so timeout 600 python tools/regimes.py --only="tokens stream" --only="headline" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/tokens_r03.log
echo "== tokens A/B again: this build"; timeout 600 python tools/regimes.py --only="tokens stream" --only="headline" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/tokens_r04.log
echo "== probe --c2"; timeout 300 ./tools/hbm_probe.bin 1000 --c2 2>&1 | tee gpurun_out/hbm_probe_c2.log | tail -20
echo "== c2 sweep"; timeout 900 python tools/c2_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c2_sweep.log | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.rstrip()); continue
    d = json.loads(l)
    print('%-62s ow=%d timing=%d  wall %6.1f us (best %6.1f)  kernel %s  %s' % (d['config'], d['overwrite'], d['hip_event_timing'], d.get('wall_us_per_step', -1), d.get('wall_us_best', -1), ('%6.1f us' % d['kernel_us']) if d.get('kernel_us') else '   -   ', d.get('error', '') or ('ok' if d.get('counters_equal') else 'COUNTERS DIFFER')))
"
echo "== quick probe"; timeout 300 ./tools/hbm_probe.bin 10000 --quick 2>&1 | tail -2
