#!/bin/bash
# round 4: key-based scan (13 instead of 16 instructions per register), saturating pack + packed domain check -- against the library before them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 --tb=short -k "sort or tiny or 1_2_4 or token or kernel_variant or every_kernel or short_and_mid or ragged or unaligned or domain or clamp or error" 2>&1 | tail -12
echo "== fuzz"; timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=line 2>&1 | tail -3
SEL='--only=tiny --only=N=48 --only=N=64 --only=N=30 --only=N=61 --only=N=7'
for i in 1 2; do
echo "== regimes, new ($i)"; timeout 600 python tools/regimes.py $SEL 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_scan_new$i.log
echo "== regimes, before ($i)"; SCV_LIB_PATH=$R/tools/ab/libscvote_r04b.so timeout 600 python tools/regimes.py $SEL 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_scan_old$i.log
done
