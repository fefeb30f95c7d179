set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fuzz seeds 500..2499"; SCV_FUZZ_FIRST=500 SCV_FUZZ_SEEDS=2000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
echo "== N=1024 dispatch"; for n in 896 900 1000 1001 1024; do echo "N=$n $(timeout 120 python tools/one_case.py --P 50000 --B 4 --N $n --rounds 3 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"; done
