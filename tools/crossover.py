#!/usr/bin/env python3
"""Where should auto-dispatch switch kernels?  Fixed total votes, N swept over 4 decades; every
path/geometry timed interleaved in one process (cold: total footprint 4 GB >> Infinity Cache)."""
from __future__ import annotations
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    eng = Engine(device=0, timing=True)
    dev = torch.device("cuda:0")
    total = 1 << 30
    B = 4
    dist = int(os.environ.get("DIST", "1"))
    variants = [("small", {"path": 3}, None)]
    for (c, t, w, u) in [(4, 256, 8, 4), (4, 256, 8, 2), (4, 256, 6, 4), (8, 256, 4, 4), (8, 256, 4, 2), (8, 512, 2, 4), (16, 512, 2, 4), (16, 1024, 1, 4)]:
        variants.append((f"stream R{c} T{t} wg{w} U{u}", {"path": 1}, (c, t, w, u)))
    out = []
    for N in [int(x) for x in os.environ.get("NS", "8,16,64,256,1024,2048,4096,8192,16384,32768,65536,131072,524288").split(",")]:
        P = max(1, total // (N * B))
        a = torch.empty((P, B, N), dtype=torch.int32, device=dev)
        tr = torch.empty((P,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(a, None, tr, P=P, B=B, N=N, seed=5, dist=dist)
        counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
        cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
        eng.sync(); eng.drain_kernel_ns()
        times = {v[0]: [] for v in variants}
        for r in range(4):
            for name, opts, tune in variants:
                if name == "small" and N > 16384:
                    continue
                for k, v in opts.items():
                    eng.set_option(k, v)
                if tune:
                    eng.set_tuning(*tune)
                counters.zero_()
                eng.aggregate_device(a, tr, counters=counters, cells=cells)
                eng.sync()
                ns, n = eng.drain_kernel_ns()
                if r:
                    times[name].append(ns / n)
        row = {"N": N, "P": P}
        for name in times:
            if times[name]:
                row[name] = P * B * N * 4 / statistics.median(times[name])
        out.append(row)
        best = max((v, k) for k, v in row.items() if k not in ("N", "P"))
        print(f"N={N:7d} P={P:8d} " + "  ".join(f"{k.replace('stream ', '')}={v:6.0f}" for k, v in row.items() if k not in ("N", "P")) + f"   best: {best[1]}", flush=True)
        del a, cells
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/crossover_d{dist}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
