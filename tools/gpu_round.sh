#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, variant sweep, rocprof summaries. Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== torch-free load"; timeout 120 python -c "import ctypes; L=ctypes.CDLL('o1_inference_scaling_laws_amd/csrc/libscvote.so'); print('devices visible without torch:', L.scv_device_count())" 2>&1 | tail -2
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json; tail -3 gpurun_out/bench.err
echo "== probe"; timeout 300 ./tools/hbm_probe.bin 42 2>&1 | tee gpurun_out/hbm_probe.log | tail -8
echo "== regimes"; timeout 600 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes.log | tail -14
echo "== C5 pass@k + bootstrap"; timeout 600 python tools/c5_passk_bootstrap.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
echo "== rocprof kernel-trace"; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_trace.log 2>&1; tail -2 $R/gpurun_out/prof_trace.log
echo "== rocprof pmc FETCH_SIZE"; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1; tail -2 $R/gpurun_out/prof_fetch.log
echo "== rocprof pmc WRITE_SIZE"; timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1; tail -2 $R/gpurun_out/prof_write.log
echo "== rocprof pmc LDS"; timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof_lds -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_lds.log 2>&1; tail -2 $R/gpurun_out/prof_lds.log
cd $R; find gpurun_out -name "*.csv" | head -30; du -sh gpurun_out
