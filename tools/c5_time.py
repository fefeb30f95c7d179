#!/usr/bin/env python3
"""Time the two bootstrap kernels on C5's table (P = 10^4 problems, 1000 resamples)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from o1_inference_scaling_laws_amd.engine import Engine
eng = Engine(device=0)
dev = torch.device("cuda:0")
P, B, N = 10000, 1, 1 << 12
ans = torch.empty((P, B, N), dtype=torch.int32, device=dev); tr = torch.empty((P,), dtype=torch.int32, device=dev)
eng.synth_fill_device(ans, None, tr, P=P, B=B, N=N, seed=55, dist=1)
counters, cells, _ = eng.aggregate_device(ans, tr)
eng.sync()
for lds in (0, 1):
    eng.set_option("boot_path", 0 if lds else 3)      # LDS-resident code table / global gathers
    out = eng.bootstrap_device(cells, 0, 1000, 7, 4); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); eng.bootstrap_device(cells, 0, 1000, 7, 4, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"boot_lds={lds}: median {sorted(ts)[5]:.1f} us  min {min(ts):.1f} us")
