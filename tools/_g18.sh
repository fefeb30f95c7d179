set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== sort kernel check"; timeout 1200 python tools/sort_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_sort_check6.log | grep -v "tok=1" | head -26
echo "== A/B double buffer"
for s in "3200000 8" "1600000 16" "3200000 7" "400000 8" "400000 16"; do set -- $s
  for o in "sort_db=1" "sort_db=0"; do for c in "" "--no-cells"; do
    echo "N=$2 $o $c: $(timeout 120 python tools/one_case.py --P $1 --B 4 --N $2 --rounds 4 --opt $o $c 2>&1 | grep -v amdgpu | tail -1 | cut -c1-170)"
  done; done
done
