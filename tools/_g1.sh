set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r3_pytest1.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r3_bench1.err | tee gpurun_out/r3_bench1.json | cut -c1-1500; tail -3 gpurun_out/r3_bench1.err
echo "== bench gpus 2 (no launcher)"; timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -2 | cut -c1-600
echo "== bench self-spawn gloo 2 ranks full size"; timeout 600 python bench.py --gpus 2 --share-device --backend gloo --steps 4 --warmup 1 --resident 2 --no-cpu-baseline 2>gpurun_out/r3_bench2.err | tee gpurun_out/r3_bench_2ranks_shared.json | cut -c1-900; tail -3 gpurun_out/r3_bench2.err
