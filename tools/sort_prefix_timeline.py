#!/usr/bin/env python3
"""Where a wave of scv_sort_prefix spends its cycles: the measurement build (tools/ab/libscvote_sptl.so: csrc/scvote_sort_prefix.hip under
-DSCV_SP_TIMELINE, built by `tools/build_variant.sh sptl scvote_sort_prefix -DSCV_SP_TIMELINE`) on pools of 64 / 32 votes with several budget lists."""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SCV_LIB_PATH"] = os.path.join(R, "tools", "ab", "libscvote_sptl.so")
sys.path.insert(0, R)
PH = ["wait", "rows->regs", "records", "sort", "block scans", "final scan", "loop"]


def main():
    import torch
    from o1_inference_scaling_laws_amd import _lib
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    L = _lib.load()
    L.scv_debug_sort_prefix_timeline.argtypes = [C.POINTER(C.c_uint64 * 8), C.c_int]
    eng = Engine(device=0, timing=True)
    dev = torch.device("cuda:0")
    for (P, N, nv) in [(200000, 64, [64]), (200000, 64, [32, 64]), (200000, 64, [1, 2, 4, 8, 16]), (200000, 64, [1, 2, 4, 8, 16, 32, 64]), (200000, 64, [2]), (200000, 64, [16]),
                       (200000, 32, [32]), (200000, 32, [1, 2, 4, 8, 16, 32])]:
        B = len(nv)
        pool = torch.empty((P, 1, N), dtype=torch.int32, device=dev)
        tr = torch.empty((P,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(pool, None, tr, P=P, B=1, N=N, seed=4, dist=1)
        nvt = torch.tensor(nv, dtype=torch.int32, device=dev)
        counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
        cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
        out = (C.c_uint64 * 8)()
        for rnd in range(4):
            counters.zero_()
            eng.aggregate_prefix_device(pool.view(P, N), tr, nvt, counters=counters, cells=cells)
            eng.sync()
            if rnd == 0:
                L.scv_debug_sort_prefix_timeline(C.byref(out), 1)
                eng.drain_kernel_ns()
        ns, n = eng.drain_kernel_ns()
        L.scv_debug_sort_prefix_timeline(C.byref(out), 1)
        t = list(out)
        steps = t[7] or 1
        print(f"N={N} nv={nv}: {ns / n / 1e3:6.1f} us/call (instrumented)  cycles/step " + "  ".join(f"{PH[i]} {t[i] / steps:7.0f}" for i in range(7)) + f"  total {sum(t[:7]) / steps:8.0f}", flush=True)


if __name__ == "__main__":
    main()
