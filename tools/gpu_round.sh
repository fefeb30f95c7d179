#!/bin/bash
# ONE GPU-box session = ONE consistent evidence set: the bench line FIRST, then rocprofv3 --kernel-trace --stats and the
# PMC passes of the SAME command, the read-ceiling probes, the parity suite, the other bench forms, the regimes table and the per-regime
# PMC passes -- all of the same library on the same box.  (Round 4's probes of the memory pipe and of the VALU issue rates -- hbm_probe --dma / --dmawork /
# --vmemq, valu_probe, sort_timeline -- are not repeated: profiles/r04_*.)  Outputs -> gpurun_out/ ; tools/summarize_profiles.py rNN copies the judged files.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== torch-free load"; timeout 120 python -c "import ctypes; L=ctypes.CDLL('o1_inference_scaling_laws_amd/csrc/libscvote.so'); print('devices visible without torch:', L.scv_device_count())" 2>&1 | tail -2
echo "== bench (the driver's command)"; timeout 900 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-500; tail -2 gpurun_out/bench.err
echo "== rocprof kernel-trace of the same command"; (cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/prof_trace.log 2>&1); tail -2 gpurun_out/prof_trace.log | cut -c1-300
echo "== rocprof pmc FETCH_SIZE"; (cd /tmp; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-read-ceiling --no-live-traffic --no-full-pass > $R/gpurun_out/prof_fetch.log 2>&1); tail -1 gpurun_out/prof_fetch.log | cut -c1-200
echo "== rocprof pmc WRITE_SIZE"; (cd /tmp; timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-read-ceiling --no-live-traffic --no-full-pass > $R/gpurun_out/prof_write.log 2>&1); tail -1 gpurun_out/prof_write.log | cut -c1-200
echo "== rocprof pmc LDS"; (cd /tmp; timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof_lds -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-read-ceiling --no-live-traffic --no-full-pass > $R/gpurun_out/prof_lds.log 2>&1); tail -1 gpurun_out/prof_lds.log | cut -c1-200
echo "== probe"; timeout 300 ./tools/hbm_probe.bin 10000 2>&1 | tee gpurun_out/hbm_probe.log | tail -4
echo "== probe --percu"; timeout 300 ./tools/hbm_probe.bin 1000 --percu 2>&1 | tee gpurun_out/hbm_probe_percu.log | tail -6
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --tb=short 2>&1 | tail -12 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench c5"; timeout 900 python bench.py --workload c5 --steps 5 --warmup 2 2> gpurun_out/bench_c5.err > gpurun_out/bench_c5.json; cut -c1-300 gpurun_out/bench_c5.json
echo "== bench c2 (eager / eager one launch / graph / graph x10)"
timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --cpu-baseline-seconds 0.2 2>/dev/null > gpurun_out/bench_c2.json
timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --c2-one-launch 2>/dev/null > gpurun_out/bench_c2_one_launch.json
timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --graph 2>/dev/null > gpurun_out/bench_c2_graph.json
timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --graph --graph-steps 10 2>/dev/null > gpurun_out/bench_c2_graph10.json
for f in bench_c2 bench_c2_one_launch bench_c2_graph bench_c2_graph10; do python -c "import json; d=json.load(open('gpurun_out/$f.json')); print('$f', round(d['ms_per_step']*1e3, 2), 'us/step', d['config']['launch'], 'kernel', round(d['roofline']['kernel_avg_ms']*1e3, 2), 'us')"; done
echo "== bench dists"; for d in 0 2 3 4 5; do timeout 600 python bench.py --dist $d --no-cpu-baseline --steps 6 --no-read-ceiling --no-live-traffic 2>/dev/null; done > gpurun_out/bench_dists.jsonl; python -c "
import json
for l in open('gpurun_out/bench_dists.jsonl'):
    d = json.loads(l); print(d['config']['distribution'], round(d['roofline']['achieved']), 'GB/s', '%.3e' % d['value'])"
echo "== bench tokens x3"; for i in 1 2 3; do timeout 600 python bench.py --tokens --problems-per-step 625 --no-cpu-baseline --steps 6 2>/dev/null; done > gpurun_out/bench_tokens.jsonl; python -c "
import json
for l in open('gpurun_out/bench_tokens.jsonl'):
    d = json.loads(l); print('tokens', round(d['roofline']['achieved']), 'GB/s of 8 B/vote', '%.3e' % d['value'], 'votes/s')"
echo "== bench --comm (one process, the library's communicator; contexts share the GPU)"
timeout 600 python bench.py --comm peer --gpus 2 --share-device --problems-per-step 600 --steps 6 --warmup 2 2>/dev/null > gpurun_out/bench_comm_peer_2ctx.json; cut -c1-200 gpurun_out/bench_comm_peer_2ctx.json
timeout 600 python bench.py --comm rccl --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_comm_rccl_1gpu.json; cut -c1-200 gpurun_out/bench_comm_rccl_1gpu.json
timeout 600 python bench.py --gpus 2 --share-device --backend gloo --problems-per-step 600 --steps 6 --warmup 2 2>/dev/null > gpurun_out/bench_2ranks_shared_gpu.json; cut -c1-200 gpurun_out/bench_2ranks_shared_gpu.json
python -c "
import json
for f in ('bench_comm_peer_2ctx', 'bench_comm_rccl_1gpu'):
    d = json.load(open('gpurun_out/%s.json' % f)); r = d['roofline']
    print(f, 'kernel ms per rank min/max', round(r['kernel_avg_ms_per_rank_min'], 3), round(r['kernel_avg_ms_per_rank_max'], 3), 'exposed all-reduce us', r['exposed_allreduce_us'], 'selftest words', d['config']['comm_selftest_words_per_rank'], 'create s', round(d['config']['comm_create_s'], 3))"
echo "== regimes"; timeout 1200 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes.log | tail -8
echo "== PMC per regime (sorted cells 48 / 64, register-resident 96 / 128 / 256 / 1024, dense 2048 / 8192, lane 3, few votes 1 / 4)"
SHAPES="6400000:4:1 3200000:4:2 1600000:4:4 1000000:4:3 400000:4:48 400000:4:64 300000:4:72 200000:4:96 200000:4:128 100000:4:256 50000:4:1024 40000:4:2048 10000:4:8192" timeout 1500 bash tools/prof_regimes.sh r06 > gpurun_out/prof_regimes_r06.log 2>&1; tail -3 gpurun_out/prof_regimes_r06.log
echo "== PMC of the prefix-budget kernels (budgets 1, 2, 4 ... N over one pool: scv_sort_prefix 32 / 64, scv_prefix_pool 256 / 1024 / 4096)"
MODE=prefix SHAPES="200000:1:32 200000:1:64 200000:1:128 100000:1:256 50000:1:1024 20000:1:4096" timeout 900 bash tools/prof_regimes.sh r06_prefix > gpurun_out/prof_regimes_r06_prefix.log 2>&1; tail -3 gpurun_out/prof_regimes_r06_prefix.log
echo "== prefix budgets over short pools (auto: DEVICE mode queues scv_sort_prefix and the general kernel for pools of 17 .. 64 votes)"; timeout 600 python tools/prefix_small.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefix_small.log | cut -c1-140 | tail -14
echo "== ... with the budgets promised to be powers of two (prefix_path = 5: one launch)"; timeout 600 python tools/prefix_small.py prefix_path=5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefix_small_promised.log | cut -c1-140 | sed -n 3,4p
echo "== ... the general kernels alone (prefix_path = 1: one lane per problem up to 64 votes)"; timeout 600 python tools/prefix_small.py prefix_path=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefix_small_lane.log | cut -c1-140 | sed -n 3,4p
echo "== prefix pools over the distributions D0 .. D5"; timeout 600 bash tools/prefix_dists.sh 2>&1 | tee gpurun_out/prefix_dists.log | tail -18
echo "== ranks from returning atomics against the register-resident cell kernels (dense cells of 96 .. 1024 votes, D0 .. D5)"; timeout 900 python tools/rtn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rtn_ab.log | cut -c1-200 | tail -6
echo "== dispatch thresholds re-checked on this box"; timeout 1500 python tools/crossovers.py 2>&1 | grep -v amdgpu.ids | tail -40
echo "== packed 4-byte records against 16-byte records against counters only (N = 1, 2, 4, 8)"
for s in "12800000 8 1" "5000000 19 1" "12800000 4 2" "6400000 4 4" "3200000 4 8"; do set -- $s; for m in "" "--packed" "--no-cells"; do timeout 300 python tools/one_case.py --P $1 --B $2 --N $3 --rounds 6 $m 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-18s records=%-10s %8.1f us %8.1f GB/s of votes' % (str(d['shape']), sys.argv[1] or '16-byte', d['median_us'], d['GBps']))" "$m"; done; done | tee gpurun_out/packed_records.log
echo "== TSAN build of the host code under the HIP runtime (best effort: the binding check is tests/test_host_sanitizers.py on the CPU)"; timeout 400 bash tools/tsan_host.sh > gpurun_out/tsan_host.log 2>&1; tail -4 gpurun_out/tsan_host.log
echo "== host mode"; timeout 600 python tools/host_mode_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/host_mode.log | tail -12
cd $R; du -sh gpurun_out
