#!/usr/bin/env python3
"""Sweep the split-N segment count for few-cell shapes (cold buffers)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from o1_inference_scaling_laws_amd.engine import Engine, counters_size
eng = Engine(device=0, timing=True)
dev = torch.device("cuda:0")
for (P, B, N, segs_list) in [(30, 1, 1 << 20, [0, 4, 8, 9, 16, 17, 25, 34]), (1, 1, 1 << 24, [0, 64, 128, 256, 512, 1024]),
                             (30, 8, 1 << 17, [0, 1, 2]), (8, 1, 1 << 22, [0, 16, 32, 64, 128])]:
    nbuf = max(2, int(600e6 // (P * B * N * 4)) + 1)
    bufs = []
    for i in range(nbuf):
        a = torch.empty((P, B, N), dtype=torch.int32, device=dev); tr = torch.empty((P,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(a, None, tr, P=P, B=B, N=N, seed=3 + i, dist=1)
        bufs.append((a, tr))
    counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
    cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
    out = []
    for segs in segs_list:
        eng.set_option("path", 0 if segs == 0 else (1 if segs == 1 else 2)); eng.set_option("segs", max(segs, 0) if segs > 1 else 0)
        eng.sync(); eng.drain_kernel_ns(); ts = []
        for r in range(5):
            for (a, tr) in bufs:
                counters.zero_(); eng.aggregate_device(a, tr, counters=counters, cells=cells); eng.sync()
                ns, n = eng.drain_kernel_ns()
                if r: ts.append(ns / n)
        med = statistics.median(ts)
        out.append(f"segs={segs if segs else 'auto'}: {med/1e3:.1f} us ({P*B*N*4/med:.0f} GB/s)")
    print(f"[{P},{B},{N}]  " + "  ".join(out), flush=True)
    eng.set_option("path", 0); eng.set_option("segs", 0)
