set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== dense A/B"; timeout 900 python tools/dense_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_dense_ab.log | tail -30
