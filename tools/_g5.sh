set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== sort kernel check"; timeout 1200 python tools/sort_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sort_check1.log | tail -60
