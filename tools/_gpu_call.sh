#!/bin/bash
# scratch entry point of a gpurun call (gpurun --timeout N -- 'bash tools/_gpu_call.sh'): edited per call; the evidence round is the default
cd /root/repo
bash tools/gpu_round.sh > gpurun_out/gpu_round_r06.log 2>&1
tail -5 gpurun_out/gpu_round_r06.log
