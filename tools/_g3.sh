set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest (parity, fuzz)"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/r3_pytest3.log
echo "== regimes"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_regimes_b.log | tail -60
cp gpurun_out/regimes.json gpurun_out/r3_regimes_b.json
echo "== bench tokens"; timeout 600 python bench.py --tokens --problems-per-step 625 --no-cpu-baseline --steps 6 2>/dev/null | tee gpurun_out/r3_bench_tokens.json | cut -c1-400
