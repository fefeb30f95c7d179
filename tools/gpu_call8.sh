#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== timeline"; timeout 600 python tools/sort_timeline.py 16 32 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort_timeline.log
echo "== timeline, pieces back to back"; timeout 600 python tools/sort_timeline.py --nospread 16 32 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort_timeline_nospread.log
