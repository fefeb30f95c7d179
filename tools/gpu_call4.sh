#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
echo "== regimes"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes.log | tail -40
echo "== pmc"; SHAPES="200000:4:64 100000:4:256 50000:4:1024 40000:4:2048 20000:8:4096" bash tools/prof_regimes.sh reg2 2>&1 | grep -E "^## |^.void scv::scv_reg|fractions|per wave" | grep -v "VALU 158"
