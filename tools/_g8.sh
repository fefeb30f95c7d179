set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== sort kernel check"; timeout 1200 python tools/sort_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sort_check4.log | tail -45
echo "== PMC sort kernel"; SHAPES="3200000:4:8 400000:4:64" timeout 1200 bash tools/prof_regimes.sh sort2 2>&1 | grep "^## \|fractions\|per wave\|profiled duration"
