set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/r3_pytest4.log
echo "== regimes (tiny)"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_regimes_c.log | grep "tiny\|N=64\|N=128\|reference"
cp gpurun_out/regimes.json gpurun_out/r3_regimes_c.json
echo "== per-CU probe"; timeout 300 ./tools/hbm_probe.bin 2500 --percu 2>&1 | tee gpurun_out/r3_hbm_probe_percu.log | tail -22
