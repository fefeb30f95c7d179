#!/usr/bin/env python3
"""One launch of scv_sort_prefix on the 100 MHz wall clock: when the LAST wave passes each mark, in microseconds after the FIRST wave started (measurement
build tools/ab/libscvote_spwall.so: `tools/build_variant.sh spwall scvote_sort_prefix -DSCV_SP_WALL`), beside the hipEvent time of the same launch."""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SCV_LIB_PATH"] = os.path.join(R, "tools", "ab", "libscvote_spwall.so")
sys.path.insert(0, R)
MARKS = ["last start", "classes", "image 1", "rows->regs", "sorted", "scanned", "steps done", "records out", "counters"]


def main():
    import torch
    from o1_inference_scaling_laws_amd import _lib
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    L = _lib.load()
    L.scv_debug_sort_prefix_wall.argtypes = [C.POINTER(C.c_uint64 * 16), C.c_int]
    eng = Engine(device=0, timing=True)
    dev = torch.device("cuda:0")
    shapes = []
    for P in (25000, 50000, 100000, 131072, 200000, 400000):
        shapes.append((P, 64, [1, 2, 4, 8, 16, 32, 64]))
    for P in (50000, 200000):
        shapes.append((P, 32, [1, 2, 4, 8, 16, 32]))
    shapes.append((200000, 64, [64]))
    for (P, N, nv) in shapes:
        B = len(nv)
        pool = torch.empty((P, 1, N), dtype=torch.int32, device=dev)
        tr = torch.empty((P,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(pool, None, tr, P=P, B=1, N=N, seed=4, dist=1)
        nvt = torch.tensor(nv, dtype=torch.int32, device=dev)
        counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
        cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
        out = (C.c_uint64 * 16)()
        rows = []
        for rnd in range(5):
            counters.zero_()
            eng.sync()
            L.scv_debug_sort_prefix_wall(C.byref(out), 1)
            eng.drain_kernel_ns()
            eng.aggregate_prefix_device(pool.view(P, N), tr, nvt, counters=counters, cells=cells, budgets_host=nv)
            eng.sync()
            ns, n = eng.drain_kernel_ns()
            L.scv_debug_sort_prefix_wall(C.byref(out), 0)
            t = list(out)
            rows.append((ns / max(n, 1) / 1e3, [(t[i] - t[0]) / 100.0 for i in range(1, 10)]))
        ev, marks = rows[-1]
        if P == 200000 and N == 64 and B == 7:
            print(f"    SIMD of waves 0..7 of workgroups 0, 1: {t[11]:016x} (lowest digit = wave 0); of workgroups 100, 101: {t[12]:016x}")
        print(f"P={P:7d} N={N:3d} B={B}: hipEvents {ev:6.1f} us | " + "  ".join(f"{MARKS[i]} {marks[i]:5.1f}" for i in range(9)), flush=True)


if __name__ == "__main__":
    main()
