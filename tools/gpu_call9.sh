#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== dmawork"; timeout 300 ./tools/hbm_probe.bin 1000 --dmawork 2>&1 | tee gpurun_out/hbm_probe_dmawork.log
