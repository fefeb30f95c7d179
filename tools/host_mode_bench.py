#!/usr/bin/env python3
"""PCIe-inclusive rate of SCV_MEM_HOST calls (numpy in, numpy out).  Never the headline `value`:
DESIGN.md quotes it beside the HBM-resident number.  Compares the round-1 serial staging loop with the
three-stage ingestion pipeline (pageable source through pinned bounce slots; pinned source DMA'd in place)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from o1_inference_scaling_laws_amd.engine import Engine, pinned_empty
from oracle import coracle

eng = Engine(device=0, timing=True)
P, B, N = 128, 8, 1 << 20          # 4.3 GB of votes
a, _, tr = coracle.synth_fill(P, B, N, 3, 1)
want = coracle.aggregate_mt(a[:8], tr[:8], 8)
ap = pinned_empty(a.shape, np.int32)
ap[...] = a
rows = []


def run(label, src, **opts):
    for k, v in opts.items():
        eng.set_option(k, v)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        res = eng.aggregate(src, tr)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert np.array_equal(res.cells["max_count"][:8], want["cells"]["max_count"]) and np.array_equal(res.cells["n_modes"][:8], want["cells"]["n_modes"])
    ns, n = eng.drain_kernel_ns()
    rows.append({"mode": label, "options": opts, "best_ms": best * 1e3, "GBps": a.nbytes / best / 1e9, "votes_per_s": a.size / best})
    print(f"{label:54s} {best*1e3:8.1f} ms  = {a.nbytes/best/1e9:6.1f} GB/s = {a.size/best:.3e} votes/s   acc={res.accuracy(0):.4f}", flush=True)


for th in (1, 4, 8, 16, 32):
    run(f"pipeline, pageable source, {th:2d} copy threads, 128 MB", a, copy_threads=th, stage_mb=128)
for mb in (32, 64, 256, 512):
    run(f"pipeline, pageable source, 16 copy threads, {mb} MB", a, copy_threads=16, stage_mb=mb)
run("pipeline, PINNED source (DMA in place), 128 MB", ap, copy_threads=8, stage_mb=128)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/host_mode.json", "w"), indent=1)
