#!/bin/bash
# round-5 development session 2: PMC of scv_prefix_pool v2 (pools of 256 / 1024 / 64 forced) and of scv_reg_cells at N = 96
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
MODE=prefix SHAPES="100000:1:256 50000:1:1024" bash tools/prof_regimes.sh r05_prefix_pool > gpurun_out/prof_r05_prefix_pool.log 2>&1
MODE=prefix EXTRA="--opt prefix_path=4 --opt reg_shape=16" SHAPES="200000:1:64" bash tools/prof_regimes.sh r05_prefix_pool64 > gpurun_out/prof_r05_prefix_pool64.log 2>&1
SHAPES="270000:4:96" bash tools/prof_regimes.sh r05_n96 > gpurun_out/prof_r05_n96.log 2>&1
cat gpurun_out/prof_regimes_r05_prefix_pool/summary.md gpurun_out/prof_regimes_r05_prefix_pool64/summary.md gpurun_out/prof_regimes_r05_n96/summary.md
