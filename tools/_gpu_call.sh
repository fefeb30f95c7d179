set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --tb=short -k "packed or few or fuzz or random or tiny or short_and_mid or host_small" 2>&1 | tail -6
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-72s %8.1f us %8.1f GB/s' % (' '.join(sys.argv[1:]), d['median_us'], d['GBps']))" "$@"; }
for s in "12800000 8 1" "12800000 4 2" "6400000 4 4"; do set -- $s
  run --P $1 --B $2 --N $3 --rounds 6
  run --P $1 --B $2 --N $3 --rounds 6 --packed
  run --P $1 --B $2 --N $3 --rounds 6 --no-cells
  run --P $1 --B $2 --N $3 --rounds 6 --no-cells --tokens
done
