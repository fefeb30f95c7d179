#!/bin/bash
# round 4, fourth GPU session: cells of exactly 1 / 2 / 4 votes (new kernel), the reverted token path, C2 bench forms
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 --tb=short -k "1_2_4 or tiny or token or kernel_variant or overwrite or reference_family or every_kernel or short_and_mid" 2>&1 | tail -25
echo "== fuzz"; timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=line 2>&1 | tail -4
echo "== regimes N=1,2,4 + tokens + C2"; timeout 600 python tools/regimes.py --only="N=1 " --only="N=2 " --only="N=4 " --only="tokens stream" --only="headline" --only="C2" --only="reference family" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_few.log
echo "== regimes N=1,2,4 with the round-3 library"; SCV_LIB_PATH=$R/tools/ab/libscvote_r03.so timeout 600 python tools/regimes.py --only="N=1 " --only="N=2 " --only="N=4 " --only="reference family" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/regimes_few_r03.log
echo "== bench c2 forms"
for f in "" "--no-timing" "--c2-one-launch" "--c2-one-launch --no-timing" "--graph" "--graph --graph-steps 10" "--graph --graph-steps 10 --c2-one-launch"; do
  timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 %-44s %.2f us/step  %.3e votes/s  launch=%s' % ('$f', d['ms_per_step']*1e3, d['value'], d['config']['launch']))"
done | tee gpurun_out/bench_c2_forms.log
