set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -x -k "sorted or sort_cells or random or short_and_mid or packed" 2>&1 | tail -6
python tools/regimes.py --only="small N=36" --only="small N=40" --only="small N=44" --only="small N=48 P=400k" --only="N=45" --only="tiny N=32 P=800k" 2>&1 | grep -v amdgpu.ids
