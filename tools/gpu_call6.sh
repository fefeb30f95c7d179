#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "split_n or overwrite or hipgraph or C2" 2>&1 | tail -2
echo "== bench c2"; timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c2 ms_per_step', d['ms_per_step'], 'kernel_avg_ms', d['roofline']['kernel_avg_ms'], 'achieved', d['roofline']['achieved'])"
timeout 600 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c2 graph ms_per_step', d['ms_per_step'])"
