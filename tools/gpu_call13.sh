#!/bin/bash
# round 4: every v_cndmask_b32_e32 (VOP2, mask in vcc: ~16 cycles per issue on gfx950) re-encoded as VOP3 (~4.6) -- A/B against the same source built the usual way
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== tests with the re-encoded library"; SCV_LIB_PATH=$R/tools/ab/libscvote_e64.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 --tb=short -x > gpurun_out/pytest_e64.log 2>&1; tail -3 gpurun_out/pytest_e64.log
for i in 1 2; do
echo "== regimes, VOP3 cndmask ($i)"; SCV_LIB_PATH=$R/tools/ab/libscvote_e64.so timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/regimes_e64_new$i.log
echo "== regimes, as compiled ($i)"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/regimes_e64_old$i.log
done
