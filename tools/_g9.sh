set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for s in "3200000 8" "1600000 16" "800000 32" "400000 64"; do set -- $s
  for o in "plain_loads=0" "plain_loads=1" "sort_kb=2"; do
    timeout 120 python tools/one_case.py --P $1 --B 4 --N $2 --rounds 4 --opt $o 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
  done
done
