#!/bin/bash
# round 4, first GPU session: the whole parity suite after the prune, smoke, the bench line (with dropin_loop), the regimes table.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --tb=short -x -k "not fuzz" 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
echo "== fuzz"; timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=line 2>&1 | tail -15 | tee gpurun_out/pytest_fuzz.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-400; tail -2 gpurun_out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print(json.dumps(d["cpu_baseline"].get("dropin_loop"), indent=1)[:3000])
PY
echo "== regimes"; timeout 1200 python tools/regimes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes.log | tail -90
du -sh gpurun_out
