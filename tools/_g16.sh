set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest subset"; timeout 1500 python -m pytest tests -m gpu -x -q -k "register_resident or not_16_byte or auto_dispatch or fuzz or prefix or ragged or sorted_cells" 2>&1 | grep -E "passed|failed" | tail -3
for s in "50000 4 900" "50000 4 1000" "50000 4 1001" "50000 4 1024" "18000 8 4501" "20000 8 3000" "40000 4 1500"; do set -- $s; echo "N=$3 $(timeout 120 python tools/one_case.py --P $1 --B $2 --N $3 --rounds 3 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"; done
