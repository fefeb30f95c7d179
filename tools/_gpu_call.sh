set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q --maxfail=10 --tb=short 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
