#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== regimes"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes.log | tail -40
