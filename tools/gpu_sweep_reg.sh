#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefix or golden or tiny" 2>&1 | tail -3
timeout 600 env SCV_FUZZ_SEEDS=500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
python tools/prefix_small.py 2>&1 | grep -v amdgpu
SHAPES="200000:4:64 100000:4:256 50000:4:1024 40000:4:2048 20000:8:4096" bash tools/prof_regimes.sh reg4 > gpurun_out/prof_reg4.log 2>&1
