#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q -k "register_resident or fuzz or golden or small_n or auto_dispatch" 2>&1 | tail -5
echo "== regimes"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | grep -E "N=|tiny|small|mid" | tee gpurun_out/regimes.log
echo "== pmc"; SHAPES="200000:4:64 100000:4:256 50000:4:1024 20000:8:4096" bash tools/prof_regimes.sh reg1 2>&1 | grep -E "^## |^.void scv::scv_reg|fractions|per wave" 
