#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
timeout 300 ./tools/valu_probe.bin > gpurun_out/valu_probe.log 2>&1; grep -E "cndmask|v_cmp|addc|v_xor" gpurun_out/valu_probe.log
