#!/bin/bash
# scratch entry point of a gpurun call (gpurun --timeout N -- 'bash tools/_gpu_call.sh'): edited per call
cd /root/repo
MODE=prefix EXTRA="--tokens --opt prefix_path=5" SHAPES="200000:1:128 200000:1:64 200000:1:32" timeout 900 bash tools/prof_regimes.sh r06_prefix_tokens > gpurun_out/prof_regimes_r06_prefix_tokens.log 2>&1
tail -40 gpurun_out/prof_regimes_r06_prefix_tokens.log
