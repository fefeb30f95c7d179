#!/bin/bash
# round-2 GPU call 1: read-ceiling probe variants, FETCH_SIZE calibration on a known-bytes kernel, baseline PMC of the short-cell kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== probe"; timeout 300 ./tools/hbm_probe.bin 10000 2>&1 | tee gpurun_out/r02_hbm_probe.log
echo "== calib FETCH_SIZE"; cd /tmp; timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_calib_fetch -- $R/tools/hbm_probe.bin 10000 --calib > $R/gpurun_out/prof_calib_fetch.log 2>&1; tail -2 $R/gpurun_out/prof_calib_fetch.log
echo "== calib WRITE_SIZE"; timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_calib_write -- $R/tools/hbm_probe.bin 10000 --calib > $R/gpurun_out/prof_calib_write.log 2>&1; tail -2 $R/gpurun_out/prof_calib_write.log
cd $R
grep -h read_cells_pipe gpurun_out/prof_calib_fetch/*/*counter_collection.csv | cut -d, -f9,16,17 | head
echo "== regimes PMC"; bash tools/prof_regimes.sh base 2>&1 | tail -150
