#!/usr/bin/env python3
"""Where a wave of scv_sort_cells spends its cycles: runs the measurement build of the library (tools/ab/libscvote_timeline.so:
csrc/scvote_sort.hip compiled with -DSCV_SORT_TIMELINE, see tools/build_timeline.sh) on the short-cell regimes and prints, per case,
the share of the wave cycles in each phase of a step and the cycles per step.  The launch time printed is NOT the kernel's: every wave
ends with eight same-address atomics.  Usage: sort_timeline.py [--nospread] [N ...]"""
from __future__ import annotations

import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = "_nospread" if "--nospread" in sys.argv else ""
os.environ["SCV_LIB_PATH"] = os.path.join(R, "tools", "ab", f"libscvote_timeline{VARIANT}.so")
sys.path.insert(0, R)

PHASES = ["wait for the copy", "rows LDS -> packed registers", "sort (+ next copy's pieces)", "scan", "records + counters", "loop control / prologue" + (" + the copy's pieces" if VARIANT else "")]


def main():
    import torch
    from o1_inference_scaling_laws_amd import _lib
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    L = _lib.load()
    L.scv_debug_sort_timeline.argtypes = [C.POINTER(C.c_uint64 * 8), C.c_int]
    eng = Engine(device=0, timing=True)
    dev = torch.device("cuda:0")
    sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [8, 16, 32, 48, 64]
    for N in sizes:
        for want_cells in (True, False):
            P = (410_000_000 // (4 * 4 * N)) // 64 * 64
            B = 4
            bufs = []
            for i in range(3):
                a = torch.empty((P, B, N), dtype=torch.int32, device=dev)
                tr = torch.empty((P,), dtype=torch.int32, device=dev)
                eng.synth_fill_device(a, None, tr, P=P, B=B, N=N, seed=11 + i, dist=1)
                bufs.append((a, tr))
            counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
            cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev) if want_cells else False
            out = (C.c_uint64 * 8)()
            for rnd in range(3):
                for (a, tr) in bufs:
                    counters.zero_()
                    eng.aggregate_device(a, tr, counters=counters, cells=cells)
                eng.sync()
                if rnd == 0:
                    L.scv_debug_sort_timeline(C.byref(out), 1)       # warm-up round: cleared
                    eng.drain_kernel_ns()
            ns, n = eng.drain_kernel_ns()
            L.scv_debug_sort_timeline(C.byref(out), 1)
            t = list(out)
            total = sum(t[:6]) or 1
            steps, waves = t[6] or 1, t[7] or 1
            print(f"N={N:3d} cells={'yes' if want_cells else 'no ':3s} {ns / n / 1e3:7.1f} us/launch (instrumented)  {total / steps:8.0f} cycles/step  "
                  f"{steps / waves:5.1f} steps/wave  " + "  ".join(f"{PHASES[i].split(' (')[0]}: {t[i] / total:.3f}" for i in range(6)), flush=True)


if __name__ == "__main__":
    main()
