cd $GRAFT_REPO_ROOT; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "tsan or no_cpp_exception" 2>&1 | tail -5
