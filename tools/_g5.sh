set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== sort kernel check"; timeout 1200 python tools/sort_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sort_check3.log | tail -60
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r3_pytest5.log
