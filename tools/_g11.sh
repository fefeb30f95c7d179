set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== dense A/B"; timeout 900 python tools/dense_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_dense_ab3.log | tail -9
for s in "20000 8 4096" "40000 4 2048" "20000 8 3000" "40000 4 1500" "20000 8 4093"; do set -- $s; timeout 120 python tools/one_case.py --P $1 --B $2 --N $3 --rounds 3 2>&1 | grep -v amdgpu | tail -1 | cut -c1-160; done
echo "== pytest subset"; timeout 1500 python -m pytest tests -m gpu -x -q -k "register_resident or not_16_byte or auto_dispatch or fuzz or prefix or ragged" 2>&1 | grep -E "passed|failed" | tail -3
