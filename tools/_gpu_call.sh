set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do
  for lib in new r05; do
    if [ $lib = r05 ]; then export SCV_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libscvote_r05.so; else unset SCV_LIB_PATH; fi
    timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-pass --no-live-traffic --no-read-ceiling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', round(d['ms_per_step'],4), 'ms/step  kernel', round(r['kernel_avg_ms'],4), 'ms', round(r['achieved']), 'GB/s')"
  done
done
