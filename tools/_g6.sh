set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest (sort kernel tests)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sorted_cells or tiny_cells or not_16_byte" 2>&1 | tail -8
echo "== PMC sort kernel"; SHAPES="3200000:4:8 1600000:4:16 800000:4:32 400000:4:64" timeout 1200 bash tools/prof_regimes.sh sort1 2>&1 | tail -120
