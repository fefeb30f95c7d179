set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== full gpu suite"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --tb=short 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== prefix_small (auto)"; timeout 600 python tools/prefix_small.py 2>&1 | grep -v amdgpu.ids | cut -c1-170 | grep "128\]" 
echo "== prefix_small promised"; timeout 600 python tools/prefix_small.py prefix_path=5 2>&1 | grep -v amdgpu.ids | cut -c1-170 | grep "128\]" 
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-72s %8.1f us %8.1f GB/s' % (' '.join(sys.argv[1:]), d['median_us'], d['GBps']))" "$@"; }
echo "== N = 72 .. 128 new / forced 1602"
for N in 72 96 100 128; do
  run --P 200000 --B 4 --N $N --rounds 6
  run --P 200000 --B 4 --N $N --rounds 6 --opt reg_shape=1602
done
run --P 200000 --B 4 --N 96 --rounds 6 --tokens
run --P 200000 --B 4 --N 96 --rounds 6 --tokens --opt reg_shape=1602
echo "== few huge cells (one-launch split-N)"
python tools/regimes.py --only="P=1 B=1 N=2^24" --only="P=30 B=1" --only="C2 30x8" --only="C2 + tokens" 2>&1 | grep -v amdgpu.ids
for segs in 64 128 256 512; do run --P 1 --B 1 --N 16777216 --rounds 6 --opt segs=$segs --opt path=2; done
for segs in 4 8 16; do run --P 30 --B 1 --N 1048576 --rounds 6 --opt segs=$segs --opt path=2; done
for segs in 2 4; do run --P 30 --B 8 --N 131072 --rounds 6 --opt segs=$segs --opt path=2; done
SCV_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libscvote_r05.so python tools/regimes.py --only="P=1 B=1 N=2^24" --only="P=30 B=1" --only="C2 30x8" 2>&1 | grep -v amdgpu.ids
