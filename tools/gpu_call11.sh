#!/bin/bash
# round 4: scv_few_votes with the truth of a step loaded with its votes (counters-only launches) -- against the library before it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 --tb=short -k "1_2_4 or sorted_cells_edges or reference_family" > gpurun_out/pytest_few.log 2>&1; tail -3 gpurun_out/pytest_few.log
echo "== fuzz"; timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=line > gpurun_out/pytest_fuzz.log 2>&1; tail -2 gpurun_out/pytest_fuzz.log
for i in 1 2; do
echo "== regimes, new ($i)"; timeout 600 python tools/regimes.py "--only=N=1 P" "--only=N=2 P" "--only=N=4 P" 2>&1 | grep -v amdgpu.ids > gpurun_out/regimes_few_new$i.log
echo "== regimes, before ($i)"; SCV_LIB_PATH=$R/tools/ab/libscvote_r04b.so timeout 600 python tools/regimes.py "--only=N=1 P" "--only=N=2 P" "--only=N=4 P" 2>&1 | grep -v amdgpu.ids > gpurun_out/regimes_few_old$i.log
done
