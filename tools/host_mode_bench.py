#!/usr/bin/env python3
"""PCIe-inclusive rate of SCV_MEM_HOST calls (numpy in, numpy out).  Never the headline `value`:
DESIGN.md quotes it beside the HBM-resident number."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from o1_inference_scaling_laws_amd.engine import Engine
from oracle import coracle

eng = Engine(device=0, timing=True)
P, B, N = 128, 8, 1 << 20          # 4.3 GB of votes
a, _, tr = coracle.synth_fill(P, B, N, 3, 1)
for pin in (0, 1, 0, 1):
    eng.set_option("pin_host", pin)
    t0 = time.perf_counter()
    res = eng.aggregate(a, tr)
    dt = time.perf_counter() - t0
    ns, n = eng.drain_kernel_ns()
    print(f"pin_host={pin}: {dt*1e3:8.1f} ms end to end  = {a.nbytes/dt/1e9:6.1f} GB/s = {a.size/dt:.3e} votes/s   (kernels {ns/1e6:.2f} ms in {n} launches)  acc={res.accuracy(0):.4f}", flush=True)
