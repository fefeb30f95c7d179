#!/bin/bash
# gpu_ab.sh <name> <regime filter> ...   (on a GPU box: gpurun -- 'bash tools/gpu_ab.sh r04b "N=64" "N=48"')
# Same-box A/B of the library in the tree against an earlier build (tools/build_ab.sh <git-rev> <name> -> tools/ab/libscvote_<name>.so, loaded
# through SCV_LIB_PATH): the parity tests that match the filters' kernels first, then tools/regimes.py on the named regimes, new / before /
# new / before.  Boxes of the pool differ by 2-3 %: only numbers of ONE invocation are compared.  Output: gpurun_out/ab_<name>_{new,old}{1,2}.log,
# gpurun_out/ab_<name>_pytest.log; `python tools/gpu_ab.sh --table <name>` is not needed: the last lines print the table.
set -u
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd $R
sel=(); for f in "$@"; do sel+=("--only=$f"); done
echo "== parity (new library)"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --maxfail=10 --tb=short > gpurun_out/ab_${name}_pytest.log 2>&1; tail -2 gpurun_out/ab_${name}_pytest.log
for i in 1 2; do
  timeout 900 python tools/regimes.py "${sel[@]}" 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_${name}_new$i.log
  SCV_LIB_PATH=$R/tools/ab/libscvote_$name.so timeout 900 python tools/regimes.py "${sel[@]}" 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_${name}_old$i.log
done
python - "$name" <<'PY'
import re, sys
name = sys.argv[1]
def load(f):
    d = {}
    for line in open(f):
        m = re.match(r"(.+?)\s+\[(\d+), (\d+), (\d+)\]\s+tok=(\d)\s+([\d.]+) us", line)
        if m:
            d[m.group(1).strip()] = float(m.group(6))
    return d
n1, n2, o1, o2 = [load(f"gpurun_out/ab_{name}_{k}.log") for k in ("new1", "new2", "old1", "old2")]
for k in n1:
    new, old = min(n1[k], n2.get(k, 9e9)), min(o1.get(k, 9e9), o2.get(k, 9e9))
    print(f"{k:44s} new {new:8.1f} us   before ({name}) {old:8.1f} us   {100 * (old / new - 1):+5.1f} %")
PY
