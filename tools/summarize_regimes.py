#!/usr/bin/env python3
"""Per-kernel PMC averages of tools/prof_regimes.sh -> markdown (copied into profiles/ by hand).
SQ_* cycle counters are in quad-cycles summed over the chip (MI355X_MICROARCH.md, cycle constants)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
print(f"# PMC summary of the short-cell regimes ({os.path.basename(root)})\n")
for shape in sorted(glob.glob(os.path.join(root, "N*")), key=lambda p: int(os.path.basename(p)[1:])):
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(shape, "g*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "scv_" not in k or "synth" in k:
                continue
            per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"## {os.path.basename(shape)}\n")
    for k, ctr in per_kernel.items():
        avg = {c: sum(v) / len(v) for c, v in ctr.items()}
        d = sorted(dur[k])[len(dur[k]) // 2]
        print(f"`{k[:90]}` (median profiled duration {d / 1e3:.1f} us)\n")
        print("| counter | avg per launch |")
        print("|---|---|")
        for c in sorted(avg):
            print(f"| {c} | {avg[c]:.6g} |")
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            parts = [f"{n} {avg[c] / wc:.2f}" for n, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"),
                                                           ("active_any", "SQ_ACTIVE_INST_ANY"), ("active_valu", "SQ_ACTIVE_INST_VALU"),
                                                           ("active_lds", "SQ_ACTIVE_INST_LDS")) if c in avg]
            print("\nfractions of SQ_WAVE_CYCLES: " + ", ".join(parts))
        if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg:
            print(f"\nper wave: VALU {avg['SQ_INSTS_VALU'] / avg['SQ_WAVES']:.0f}, LDS {avg.get('SQ_INSTS_LDS', 0) / avg['SQ_WAVES']:.0f}, "
                  f"VMEM_RD {avg.get('SQ_INSTS_VMEM_RD', 0) / avg['SQ_WAVES']:.0f}, SALU {avg.get('SQ_INSTS_SALU', 0) / avg['SQ_WAVES']:.0f} instructions")
        print()
