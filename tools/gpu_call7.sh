#!/bin/bash
# round 4, seventh GPU session: record stores left in flight across the top-of-step wait of scv_sort_cells (vmcnt(late)); phase timeline of the sort kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 --tb=short -k "sort or tiny or 1_2_4 or token or kernel_variant or every_kernel or short_and_mid or ragged or unaligned" 2>&1 | tail -8
echo "== fuzz"; timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=line 2>&1 | tail -3
SEL='--only=tiny --only=N=48 --only=N=64 --only=N=30 --only=N=61 --only=N=7'
for i in 1 2; do
echo "== regimes, new ($i)"; timeout 600 python tools/regimes.py $SEL 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_late_new$i.log
echo "== regimes, before ($i)"; SCV_LIB_PATH=$R/tools/ab/libscvote_r04a.so timeout 600 python tools/regimes.py $SEL 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_late_old$i.log
done
echo "== timeline"; timeout 600 python tools/sort_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort_timeline.log
