set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for s in "40000 4 2048" "20000 8 4096" "40000 4 1500"; do set -- $s
  echo "new: $(timeout 120 python tools/one_case.py --P $1 --B $2 --N $3 --rounds 4 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150)"
  echo "old: $(SCV_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_tmp/libscvote_olddense.so timeout 120 python tools/one_case.py --P $1 --B $2 --N $3 --rounds 4 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150)"
done; done
