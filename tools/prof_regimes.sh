#!/bin/bash
# PMC passes for the short-cell regimes ("put PMC summaries under profiles/ first").
# One rocprofv3 --pmc run per counter group per shape (no trace domains combined with --pmc).
# Usage: tools/prof_regimes.sh <tag>   -> gpurun_out/prof_regimes_<tag>/<shape>/<group>/...
#   SHAPES="P:B:N ..."  DIST=d  MODE=prefix (budgets 1, 2, 4 ... N over one pool [P, N]; B ignored)  EXTRA="--opt key=value ..." (one_case.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-base}
OUT=$R/gpurun_out/prof_regimes_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
G3="FETCH_SIZE GRBM_GUI_ACTIVE"
SHAPES=${SHAPES:-"400000:4:8 200000:4:64 100000:4:256 50000:4:1024 40000:4:2048 20000:8:4096"}
MODEFLAG=""; if [ "${MODE:-}" = "prefix" ]; then MODEFLAG="--prefix"; fi
for s in $SHAPES; do
  P=${s%%:*}; rest=${s#*:}; B=${rest%%:*}; N=${rest#*:}
  i=0
  for g in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    d=$OUT/N$N/g$i
    mkdir -p $d
    timeout 300 rocprofv3 --pmc $g --output-format csv -d $d -- python $R/tools/one_case.py --P $P --B $B --N $N --rounds 2 --dist ${DIST:-1} $MODEFLAG ${EXTRA:-} > $d/run.log 2>&1
    tail -1 $d/run.log | cut -c1-200
  done
done
python $R/tools/summarize_regimes.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
