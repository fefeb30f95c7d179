#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, variant sweep, rocprof summary. Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json; tail -5 gpurun_out/bench.err
echo "== sweep"; timeout 900 python tools/sweep.py --out gpurun_out/sweep.json 2>&1 | tee gpurun_out/sweep.log | tail -40
