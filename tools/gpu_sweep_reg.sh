#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
R=$(pwd)
for sh in 1601 1604; do
  d=$R/gpurun_out/kt_$sh; rm -rf $d; mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/tools/one_case.py --P 200000 --B 4 --N 64 --rounds 3 --opt reg_shape=$sh > $d/run.log 2>&1)
  echo "shape $sh"; python - <<PY
import csv,glob
for f in glob.glob("$d/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "scv_reg" in r["Name"] or "reduce" in r["Name"]:
            print("  ", r["Name"][:60], r["Calls"], "avg_us", float(r["AverageNs"])/1e3, "min_us", float(r["MinNs"])/1e3)
PY
done
run() { python tools/one_case.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('%-22s %-40s %8.1f us %8.1f GB/s' % (r['shape'], r['opts'], r['median_us'], r['GBps']))"; }
run --P 200000 --B 4 --N 64 --opt fused_counters_max=1073741824
run --P 200000 --B 4 --N 64 --dist 0
run --P 200000 --B 1 --N 256
run --P 25000 --B 32 --N 64
