#!/bin/bash
# round 4, second GPU session: the rest of the parity suite, tokens through LDS-DMA (A/B against the round-3 library on the same box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== pytest -m gpu (no -x)"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -k "not fuzz" 2>&1 | tail -80 | tee gpurun_out/pytest_gpu.log
echo "== fuzz"; timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=line 2>&1 | tail -8 | tee gpurun_out/pytest_fuzz.log
echo "== tokens A/B: round-3 library"; SCV_LIB_PATH=$R/tools/ab/libscvote_r03.so timeout 600 python tools/regimes.py --only=tokens --only="headline" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tokens_r03.log
echo "== tokens A/B: this build"; timeout 600 python tools/regimes.py --only=tokens --only="headline" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tokens_r04.log
echo "== bench --tokens"; for i in 1 2 3; do timeout 600 python bench.py --tokens --problems-per-step 625 --no-cpu-baseline --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tokens run', d['roofline']['achieved'], 'GB/s', d['ms_per_step'], 'ms/step')"; done | tee gpurun_out/bench_tokens_3runs.log
echo "== probe"; timeout 300 ./tools/hbm_probe.bin 10000 --quick 2>&1 | tail -3
