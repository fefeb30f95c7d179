#!/bin/bash
# PMC passes (separate runs, no trace domains) for scv_lane_prefix on the reference-shaped workload:
# maj@1, 2, 4 ... 64 over 2*10^5 pools of 64 samples.  Output: gpurun_out/prof_lp/summary.md
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
G3="FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE"
i=0
for g in "$G1" "$G2" "$G3"; do
  i=$((i+1)); d=$R/gpurun_out/prof_lp/g$i; rm -rf $d; mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $g --output-format csv -d $d -- python $R/tools/one_case.py --prefix --P 200000 --N 64 --rounds 2 > $d/run.log 2>&1)
done
python - <<PY > $R/gpurun_out/prof_lp/summary.md
import csv, glob, collections
acc = collections.defaultdict(list)
name = None
for f in glob.glob("$R/gpurun_out/prof_lp/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lane_prefix" in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["duration_ns (under PMC collection)"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("# PMC of scv_lane_prefix: maj@1,2,4..64 over 200000 pools of 64 samples (tools/prof_prefix_lane.sh)\n")
print(f"`{name}`\n\n| counter | avg per launch |\n|---|---|")
avg = {k: sum(v) / len(v) for k, v in acc.items()}
for k, v in sorted(avg.items()):
    print(f"| {k} | {v:.6g} |")
if "SQ_WAVE_CYCLES" in avg:
    w = avg["SQ_WAVE_CYCLES"]
    print(f"\nfractions of SQ_WAVE_CYCLES: wait_any {avg['SQ_WAIT_ANY']/w:.2f}, wait_inst {avg['SQ_WAIT_INST_ANY']/w:.2f}, active_valu {avg['SQ_ACTIVE_INST_VALU']/w:.2f}")
if "SQ_WAVES" in avg and "SQ_INSTS_VALU" in avg:
    n = avg["SQ_WAVES"]
    print(f"\nper wave (64 problems): VALU {avg['SQ_INSTS_VALU']/n:.0f}, SALU {avg['SQ_INSTS_SALU']/n:.0f}, LDS {avg['SQ_INSTS_LDS']/n:.0f}, VMEM_RD {avg['SQ_INSTS_VMEM_RD']/n:.0f}, VMEM_WR {avg['SQ_INSTS_VMEM_WR']/n:.0f} instructions")
if "FETCH_SIZE" in avg:
    print(f"\nHBM read = FETCH_SIZE x 1024 x 2 (gfx950) = {avg['FETCH_SIZE']*2048/1e6:.1f} MB (pool: 51.2 MB); written = WRITE_SIZE x 1024 = {avg.get('WRITE_SIZE',0)*1024/1e6:.1f} MB (cells: 22.4 MB)")
PY
cat $R/gpurun_out/prof_lp/summary.md
