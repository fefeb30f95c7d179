#!/usr/bin/env python3
"""Run ONE (P, B, N) shape of the hot path a few times on cold buffers -- the unit rocprofv3 wraps for
per-regime PMC passes (tools/prof_regimes.sh).  Prints one JSON line with the hipEvent timing."""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_prefix(eng, torch, P, N, tokens, dist, rounds, nv_list=None, no_cells=False):
    import statistics
    from o1_inference_scaling_laws_amd.engine import counters_size
    dev = torch.device("cuda:0")
    nv = nv_list or [1 << k for k in range(N.bit_length()) if (1 << k) <= N]
    B = len(nv)
    pool = torch.empty((P, 1, N), dtype=torch.int32, device=dev)
    tk = torch.empty((P, 1, N), dtype=torch.int32, device=dev) if tokens else None
    tr = torch.empty((P,), dtype=torch.int32, device=dev)
    eng.synth_fill_device(pool, tk, tr, P=P, B=1, N=N, seed=4, dist=dist)
    nvt = torch.tensor(nv, dtype=torch.int32, device=dev)
    counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
    cells = None if no_cells else torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
    eng.sync(); eng.drain_kernel_ns()
    ts = []
    for r in range(rounds + 1):
        counters.zero_()
        eng.aggregate_prefix_device(pool.view(P, N), tr, nvt, tokens=None if tk is None else tk.view(P, N), counters=counters, cells=cells)
        eng.sync()
        ns, n = eng.drain_kernel_ns()
        if r:
            ts.append(ns / n)
    med = statistics.median(ts)
    return {"shape": [P, B, N], "prefix": True, "median_us": med / 1e3, "GBps": P * N * 4 / med, "votes_per_s": P * sum(nv) / (med * 1e-9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, required=True)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--N", type=int, required=True)
    ap.add_argument("--tokens", action="store_true")
    ap.add_argument("--dist", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (scv_set_option)")
    ap.add_argument("--no-cells", action="store_true", help="counters only: no cell table is written")
    ap.add_argument("--nv", default="", help="prefix mode: the budgets, comma separated (default 1, 2, 4 ... N)")
    ap.add_argument("--prefix", action="store_true", help="prefix budgets 1, 2, 4 ... N over one pool [P, N] (B is ignored)")
    ap.add_argument("--packed", action="store_true", help="SCV_FLAG_PACKED_CELLS: 4-byte cell records (N <= 127)")
    args = ap.parse_args()
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from regimes import run
    eng = Engine(device=0, timing=True, packed_cells=args.packed)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    if args.prefix:
        r = run_prefix(eng, torch, args.P, args.N, args.tokens, args.dist, args.rounds, [int(x) for x in args.nv.split(',')] if args.nv else None, args.no_cells)
    else:
        r = run(eng, torch, args.P, args.B, args.N, args.tokens, dist=args.dist, rounds=args.rounds, want_cells=not args.no_cells)
    r["opts"] = args.opt
    print(json.dumps(r))


if __name__ == "__main__":
    main()
