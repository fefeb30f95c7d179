set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "capturable or exactly_1_2_4 or one_vote" 2>&1 | tail -12
