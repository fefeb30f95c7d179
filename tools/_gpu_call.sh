set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -x -k "two_votes or one_vote or exactly_1_2_4 or packed or random or prefix_calls or short_and_mid" 2>&1 | tail -12
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-72s %8.1f us %8.1f GB/s' % (' '.join(sys.argv[1:]), d['median_us'], d['GBps']))" "$@"; }
for m in "" "--packed" "--no-cells" "--no-cells --tokens"; do run --P 12800000 --B 4 --N 2 --rounds 6 $m; done
for m in "" "--packed" "--no-cells"; do SCV_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libscvote_prev.so run --P 12800000 --B 4 --N 2 --rounds 6 $m; done
