#!/bin/bash
# round 4, sixth GPU session: VALU issue rates (tools/valu_probe), the one-mask epilogue of scv_reg_cells (K >= 2) against the library before it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== valu probe"; timeout 300 ./tools/valu_probe.bin 2>&1 | tee gpurun_out/valu_probe.log
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 --tb=short -k "register or reg_ or short_and_mid or kernel_variant or every_kernel or ragged or unaligned" 2>&1 | tail -8
echo "== fuzz"; timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=line 2>&1 | tail -3
for i in 1 2; do
echo "== regimes, new ($i)"; timeout 600 python tools/regimes.py --only="N=128" --only="N=96" --only="N=64 P=200000" --only="N=253" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_onemask_new$i.log
echo "== regimes, before ($i)"; SCV_LIB_PATH=$R/tools/ab/libscvote_r04a.so timeout 600 python tools/regimes.py --only="N=128" --only="N=96" --only="N=64 P=200000" --only="N=253" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_onemask_old$i.log
done
