set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== sort kernel check"; timeout 1200 python tools/sort_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sort_check6.log | grep -v "tok=1" | head -24
echo "== dense A/B"; timeout 900 python tools/dense_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_dense_ab2.log | tail -12
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | head -6
for s in "20000 8 4096" "40000 4 2048" "20000 8 3000" "40000 4 1500"; do set -- $s; timeout 120 python tools/one_case.py --P $1 --B $2 --N $3 --rounds 3 2>&1 | grep -v amdgpu | tail -1 | cut -c1-160; done
