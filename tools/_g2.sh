set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest (register kernels, fuzz, unaligned)"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/r3_pytest2.log
echo "== pivots A/B"; timeout 900 python tools/pivots_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_pivots_ab.log
echo "== regimes"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_regimes_a.log | tail -50
cp gpurun_out/regimes.json gpurun_out/r3_regimes_a.json
