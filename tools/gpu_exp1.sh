#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== probe"; timeout 300 ./tools/hbm_probe.bin 42 2>&1 | tee gpurun_out/hbm_probe.log
echo "== sweep stagger/plain"; timeout 900 python tools/sweep.py --problems 1250 --copies 16 --threads 1024,512 --wg 1,2 --unroll 4,8 --dists 1 --rounds 5 --balance 1 --stagger 0,4099,65536 --plain 0,1 --top 60 --out gpurun_out/sweep_stagger.json 2>&1 | tee gpurun_out/sweep_stagger.log | tail -64
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/bench.err | tee gpurun_out/bench2.json
