set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for s in "3200000 8" "1600000 16" "800000 32"; do set -- $s
  for o in "sort_waves=0" "sort_waves=32" "sort_waves=24"; do for c in "" "--no-cells"; do
    echo "N=$2 $o $c: $(timeout 120 python tools/one_case.py --P $1 --B 4 --N $2 --rounds 4 --opt $o $c 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150)"
  done; done
done
