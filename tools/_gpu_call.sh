#!/bin/bash
cd /root/repo
bash tools/gpu_round.sh > gpurun_out/gpu_round_r06.log 2>&1
tail -5 gpurun_out/gpu_round_r06.log
