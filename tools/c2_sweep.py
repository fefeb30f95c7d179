#!/usr/bin/env python3
"""BASELINE config 2 (30 x 8 cells of 2^17 votes = 125.8 MB, one evaluation = ONE launch in overwrite mode): is 240 workgroups x
512 KiB the best geometry?  Sweep the kernel over 240 / 480 / 960 work items (whole cells; split-N in 2 / 4 segments + the merge
launch) x workgroup shapes, eager, 5 distinct tensors cycled (cold Infinity Cache), with and without the library's hipEvent
timing -- wall time per step and kernel time.  tools/hbm_probe.bin --c2 is the pure-read side of the same question.
One JSON under gpurun_out/.  (VERDICT r3 next #9.)"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from o1_inference_scaling_laws_amd._lib import ScvError
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    dev = torch.device("cuda:0")
    P, B, N = 30, 8, 1 << 17
    rows = []
    configs = [  # (label, path, segs, (copies, threads, wg_per_cu, unroll) | None)
        ("auto: 240 whole cells, 1024 threads, R16", 0, 0, None),
        ("240 whole cells, 512 threads x 2 per CU, R16", 1, 0, (16, 512, 2, 4)),
        ("240 whole cells, 256 threads x 4 per CU, R8", 1, 0, (8, 256, 4, 4)),
        ("480 half cells, 1024 threads (2 rounds) + merge", 2, 2, (16, 1024, 1, 4)),
        ("480 half cells, 512 threads x 2 per CU + merge", 2, 2, (16, 512, 2, 4)),
        ("960 quarter cells, 512 threads x 2 per CU (2 rounds) + merge", 2, 4, (16, 512, 2, 4)),
        ("960 quarter cells, 256 threads x 4 per CU, R8 + merge", 2, 4, (8, 256, 4, 4)),
        ("960 quarter cells, 256 threads x 4 per CU, R16 + merge", 2, 4, (16, 256, 4, 4)),
    ]
    for timing in (True, False):
        eng = Engine(device=0, timing=timing)
        bufs = []
        for i in range(5):
            a = torch.empty((P, B, N), dtype=torch.int32, device=dev)
            tr = torch.empty((P,), dtype=torch.int32, device=dev)
            eng.synth_fill_device(a, None, tr, P=P, B=B, N=N, seed=100 + i, dist=1)
            bufs.append((a, tr))
        counters = torch.empty(counters_size(B), dtype=torch.int64, device=dev)
        cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
        ref = None
        for label, path, segs, tune in configs:
            for overwrite in (True, False):
                try:
                    eng.set_option("path", path); eng.set_option("segs", segs)
                    if tune:
                        eng.set_tuning(*tune)
                    else:
                        eng.set_tuning(-1, -1, -1, -1)
                    def step(i):
                        a, tr = bufs[i % 5]
                        if not overwrite:
                            counters.zero_()
                        eng.aggregate_device(a, tr, counters=counters, cells=cells, overwrite=overwrite)
                    for i in range(20):
                        step(i)
                    eng.sync()
                    if timing:
                        eng.drain_kernel_ns()
                    walls = []
                    for rep in range(5):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for i in range(200):
                            step(i)
                        torch.cuda.synchronize()
                        walls.append((time.perf_counter() - t0) / 200 * 1e6)
                    kern = None
                    if timing:
                        ns, n = eng.drain_kernel_ns()
                        kern = ns / max(n, 1) / 1e3
                    c = counters.cpu().numpy().copy()
                    if ref is None:
                        ref = c
                    ok = bool((c == ref).all())                          # (same last tensor in every configuration: same counters)
                    r = {"config": label, "overwrite": overwrite, "hip_event_timing": timing, "wall_us_per_step": statistics.median(walls), "wall_us_best": min(walls),
                         "kernel_us": kern, "GBps_of_wall": P * B * N * 4 / statistics.median(walls) / 1e3, "counters_equal": ok}
                except ScvError as e:
                    r = {"config": label, "overwrite": overwrite, "hip_event_timing": timing, "error": str(e)}
                rows.append(r)
                print(json.dumps(r), flush=True)
        eng.set_option("path", 0); eng.set_option("segs", 0); eng.set_tuning(-1, -1, -1, -1)
        eng.close()
        del bufs
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/c2_sweep.json", "w"), indent=1)


if __name__ == "__main__":
    main()
