#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (c3 / c5 / c2), probes, regimes, rocprofv3 + PMC passes. Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== torch-free load"; timeout 120 python -c "import ctypes; L=ctypes.CDLL('o1_inference_scaling_laws_amd/csrc/libscvote.so'); print('devices visible without torch:', L.scv_device_count())" 2>&1 | tail -2
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-600; tail -2 gpurun_out/bench.err
echo "== bench c5"; timeout 900 python bench.py --workload c5 --steps 5 --warmup 2 2> gpurun_out/bench_c5.err > gpurun_out/bench_c5.json; cut -c1-400 gpurun_out/bench_c5.json
echo "== bench c2 (eager / graph)"; timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_c2.json; timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --graph 2>/dev/null > gpurun_out/bench_c2_graph.json; timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --cpu-baseline-seconds 0.2 --graph --graph-steps 10 2>/dev/null > gpurun_out/bench_c2_graph10.json; cut -c1-300 gpurun_out/bench_c2.json gpurun_out/bench_c2_graph.json gpurun_out/bench_c2_graph10.json
echo "== bench dists"; for d in 0 2 3; do timeout 600 python bench.py --dist $d --no-cpu-baseline --steps 6 2>/dev/null; done > gpurun_out/bench_dists.jsonl; cut -c1-200 gpurun_out/bench_dists.jsonl
echo "== bench tokens"; timeout 600 python bench.py --tokens --problems-per-step 625 --no-cpu-baseline --steps 6 2>/dev/null > gpurun_out/bench_tokens.json; cut -c1-300 gpurun_out/bench_tokens.json
echo "== probe"; timeout 300 ./tools/hbm_probe.bin 10000 2>&1 | tee gpurun_out/hbm_probe.log | tail -4
echo "== regimes"; timeout 900 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes.log | tail -30
echo "== sorted cells: parity sweep + old vs new"; timeout 900 python tools/sort_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sort_check.log | grep -v "tok=1" | head -12
echo "== dense 4096 < N <= 8192"; timeout 600 python tools/dense_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dense_ab.log | tail -9
echo "== LDS-DMA path ceiling"; timeout 300 ./tools/hbm_probe.bin 1000 --dma 2>&1 | tee gpurun_out/hbm_probe_dma.log | tail -8
echo "== prefix budgets over short pools"; timeout 600 python tools/prefix_small.py 2>&1 | grep -v amdgpu.ids | tail -14
echo "== host mode"; timeout 600 python tools/host_mode_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/host_mode.log | tail -12
echo "== rocprof kernel-trace"; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_trace.log 2>&1; tail -2 $R/gpurun_out/prof_trace.log | cut -c1-300
echo "== rocprof pmc FETCH_SIZE"; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1; tail -1 $R/gpurun_out/prof_fetch.log | cut -c1-200
echo "== rocprof pmc WRITE_SIZE"; timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1; tail -1 $R/gpurun_out/prof_write.log | cut -c1-200
echo "== rocprof pmc LDS"; timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof_lds -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_lds.log 2>&1; tail -1 $R/gpurun_out/prof_lds.log | cut -c1-200
cd $R; du -sh gpurun_out
