#!/usr/bin/env python3
"""Copy the judged rocprofv3 evidence from gpurun_out/ (scratch) into profiles/ (tracked):
kernel-trace --stats summary + per-kernel PMC averages with the gfx950 FETCH_SIZE x2 correction
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Usage: summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
os.makedirs(P, exist_ok=True)


def newest(pattern, containing=None):
    """newest file matching the pattern; `containing`: only files whose text has that string (the profiled command spawns the read-ceiling
    probe, a process with trace files of its own)"""
    files = glob.glob(os.path.join(G, pattern))
    if containing:
        files = [f for f in files if containing in open(f).read()]
    return max(files, key=os.path.getmtime) if files else None


lines = [f"# rocprofv3 summary {tag} (MI355X, `python bench.py --steps 8 --warmup 2 --no-cpu-baseline`; same box and session as {tag}_bench.json: tools/gpu_round.sh)", ""]
ks = newest("prof_trace/*/*_kernel_stats.csv", "scv_hist_argmax")
if ks:
    shutil.copy(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
    lines += ["## kernel-trace --stats (8 timed + 2 warmup + 1 parity step)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks)):
        lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
    lines.append("")
kt = newest("prof_trace/*/*_kernel_trace.csv", "scv_hist_argmax")
if kt:
    for r in csv.DictReader(open(kt)):
        if "scv_hist_argmax" in r["Kernel_Name"]:
            lines += [f"Dispatch geometry of `scv_hist_argmax`: grid {r['Grid_Size_X']} threads / workgroup {r['Workgroup_Size_X']}"
                      f" = {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])} workgroups; rocprofv3's trace fields: LDS_Block_Size {r['LDS_Block_Size']} B "
                      f"(STATIC LDS only: this kernel's LDS is dynamic, sized by the host), VGPR_Count {r['VGPR_Count']} (the trace's own unit), SGPR_Count {r['SGPR_Count']}, scratch {r['Scratch_Size']}."]
            # what the kernel really occupies: the compiler's metadata (tools/kernel_resources.py) and the host's LDS request
            try:
                import re
                sys.path.insert(0, os.path.join(R, "tools"))
                import kernel_resources
                m = re.search(r"scv_hist_argmax<(\d+), (\d+), (\d+), (\w+), (\w+)>", r["Kernel_Name"])
                want = f"scv_hist_argmax<{m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}, {m.group(5)}>" if m else None
                row = next((x for x in kernel_resources.collect() if x[0] == want), None)
                if row:
                    copies = 1 << int(m.group(1))
                    lines += [f"Compiler metadata of `{want}`: {row[1]} VGPRs, {row[3]} SGPRs, {row[4]} B scratch; dynamic LDS requested by the host = "
                              f"(1024 bins x {copies} copies + 96 words) x 4 B = {(1024 * copies + 96) * 4} B per workgroup."]
            except Exception as e:      # (the summary must not fail on a box without hipcc)
                lines += [f"(compiler metadata not available here: {e})"]
            lines += [""]
            break
pmc = collections.defaultdict(list)
for sub in ("prof_fetch", "prof_write", "prof_lds"):
    f = newest(f"{sub}/*/*_counter_collection.csv")
    if not f:
        continue
    # one file per process of the profiled command (bench.py and the read-ceiling probe it spawns): take every file of
    # the newest run
    t = os.path.getmtime(f)
    for g in glob.glob(os.path.join(G, f"{sub}/*/*_counter_collection.csv")):
        if os.path.getmtime(g) < t - 120:
            continue
        for r in csv.DictReader(open(g)):
            if "scv_hist_argmax" in r["Kernel_Name"]:
                pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if pmc:
    avg = {k: sum(v) / len(v) for k, v in pmc.items()}
    lines += ["## PMC (separate --pmc passes, per launch of scv_hist_argmax, averaged)", "", "| counter | value |", "|---|---|"]
    for k, v in sorted(avg.items()):
        lines.append(f"| {k} | {v:.6g} |")
    lines.append("")
    algo = 1250 * 8 * (1 << 20) * 4
    if "FETCH_SIZE" in avg:
        fetch = avg["FETCH_SIZE"] * 1024 * 2       # KiB units; gfx950 reports 1/2 for wide coalesced streams
        lines.append(f"HBM read traffic per launch = FETCH_SIZE x 1024 B x 2 (gfx950 correction) = {fetch/1e9:.3f} GB "
                     f"vs algorithmic {algo/1e9:.3f} GB -> ratio {fetch/algo:.4f} (no wasted re-reads).")
    if "WRITE_SIZE" in avg:
        lines.append(f"HBM write traffic per launch = WRITE_SIZE x 1024 B = {avg['WRITE_SIZE']*1024/1e6:.3f} MB (cell records + counters).")
    if "SQ_LDS_IDX_ACTIVE" in avg and "SQ_LDS_BANK_CONFLICT" in avg:
        lines.append(f"LDS: bank-conflict cycles / active cycles = {avg['SQ_LDS_BANK_CONFLICT']/avg['SQ_LDS_IDX_ACTIVE']:.3f}; "
                     f"active cycles per 64-vote ds_add_u32 = {avg['SQ_LDS_IDX_ACTIVE']/(algo/4/64):.2f}.")
    json.dump({"pmc_avg": avg, "hbm_traffic_bytes": (avg.get("FETCH_SIZE", 0) * 2048 + avg.get("WRITE_SIZE", 0) * 1024)},
              open(os.path.join(P, f"{tag}_pmc.json"), "w"), indent=1)
# every other file of the same session that the documents quote (one box, one invocation: tools/gpu_round.sh)
for name in ("bench.json", "hbm_probe.log", "hbm_probe_percu.log", "hbm_probe_dma.log", "bench_c5.json", "bench_c2.json", "bench_c2_one_launch.json", "bench_c2_graph.json",
             "bench_c2_graph10.json", "bench_dists.jsonl", "bench_tokens.jsonl", "bench_comm_peer_2ctx.json", "bench_comm_rccl_1gpu.json", "bench_2ranks_shared_gpu.json",
             "regimes.log", "regimes.json", "sort_check.log", "prefix_small.log", "host_mode.log", "pytest_gpu.log", "smoke.log",
             "valu_probe.log", "sort_timeline.log", "sort_timeline_nospread.log", "hbm_probe_dmawork.log", "hbm_probe_vmemq.log",
             "crossovers.md", "packed_records.log", "tsan_host.log", "prefix_small_promised.log", "prefix_small_lane.log", "prefix_small_general.log", "prefix_dists.log", "rtn_ab.log", "sort_prefix_timeline.log", "gpu_round_" + tag + ".log"):
    src = os.path.join(G, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{name.replace('bench_c5.json', 'bench_c5_1gpu.json')}"))
for sub, out in ((f"prof_regimes_{tag}", f"{tag}_regimes_pmc.md"), (f"prof_regimes_{tag}_prefix", f"{tag}_prefix_pmc.md")):
    summ = os.path.join(G, sub, "summary.md")
    if os.path.exists(summ):
        shutil.copy(summ, os.path.join(P, out))
open(os.path.join(P, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
