#!/usr/bin/env python3
"""Latency of reference-sized prefix calls (P = 30 ... 3000 pools, budgets 1, 2, 4 ... N): blocking HOST-mode calls (the drop-in's form) and the
kernel time inside them, scv_sort_prefix (auto) against the general kernels (prefix_path = 1 / 4)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from o1_inference_scaling_laws_amd.engine import Engine
    from oracle import coracle
    eng = Engine(device=0, timing=True)
    for N in (32, 64, 128):
        for P in (30, 300, 3000, 30000):
            a, t, tr = coracle.synth_fill(P, 1, N, 5, 1, want_tokens=True)
            pool, tpool = np.ascontiguousarray(a[:, 0, :]), np.ascontiguousarray(t[:, 0, :])
            nv = np.array([1 << k for k in range(N.bit_length()) if (1 << k) <= N], dtype=np.int32)
            row = f"N={N:4d} P={P:6d} B={len(nv)}"
            for name, opts in (("auto", {}), ("lane/pool", {"prefix_path": 1 if N <= 64 else 4})):
                for k, v in opts.items():
                    eng.set_option(k, v)
                for tok in (None, tpool):
                    for _ in range(5):
                        eng.aggregate_prefix(pool, tr, nv, tokens=tok)
                    eng.drain_kernel_ns()
                    ts = []
                    for _ in range(30):
                        t0 = time.perf_counter()
                        eng.aggregate_prefix(pool, tr, nv, tokens=tok)
                        ts.append(time.perf_counter() - t0)
                    ns, n = eng.drain_kernel_ns()
                    row += f" | {name}{' +tok' if tok is not None else ''}: call {statistics.median(ts) * 1e6:6.1f} us, kernel {ns / 30 / 1e3:5.1f}"
                for k in opts:
                    eng.set_option(k, 0)
            print(row, flush=True)


if __name__ == "__main__":
    main()
