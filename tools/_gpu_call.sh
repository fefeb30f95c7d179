#!/bin/bash
# scratch entry point of a gpurun call (gpurun --timeout N -- 'bash tools/_gpu_call.sh'): edited per call
cd /root/repo
bash tools/_driver_like.sh 2>&1 | grep -v amdgpu.ids
