set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for s in "50000 4 1024" "60000 4 768" "100000 4 512" "100000 4 256"; do set -- $s
  for o in "reg_shape=0" "reg_shape=1041" "reg_shape=6404" "reg_shape=3204"; do
    echo "$o: $(timeout 120 python tools/one_case.py --P $1 --B $2 --N $3 --rounds 3 --opt $o 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"
  done
done
for d in 0 3; do echo "dist $d N=1024 dense: $(timeout 120 python tools/one_case.py --P 50000 --B 4 --N 1024 --rounds 3 --dist $d --opt reg_shape=1041 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"; done
