#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-1500; tail -3 gpurun_out/bench.err
echo "== bench c5"; timeout 900 python bench.py --workload c5 --steps 5 --warmup 2 2> gpurun_out/bench_c5.err | tee gpurun_out/bench_c5.json | cut -c1-1800; tail -3 gpurun_out/bench_c5.err
