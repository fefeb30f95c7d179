#!/usr/bin/env python3
"""Within-process interleaved A/B of the hot-path kernel variants (copies x threads x wg/CU x unroll
x distribution) on one resident slab.  Reports median / min kernel time from the library's hipEvents
and the algorithmic HBM GB/s.  GPU box only:  python tools/sweep.py --out gpurun_out/sweep.json"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=256)
    ap.add_argument("--budgets", type=int, default=8)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--copies", default="8,16,32")
    ap.add_argument("--threads", default="256,512,1024")
    ap.add_argument("--wg", default="1,2,4")
    ap.add_argument("--unroll", default="2,4,8")
    ap.add_argument("--dists", default="1,0,2")
    ap.add_argument("--tokens", action="store_true")
    ap.add_argument("--balance", default="1", help="comma list of 0/1: balanced-grid option")
    ap.add_argument("--grids", default="0", help="comma list of explicit grid sizes (0 = derive)")
    ap.add_argument("--stagger", default="0", help="comma list of stagger_vecs values")
    ap.add_argument("--plain", default="0", help="comma list of 0/1: plain instead of non-temporal loads")
    ap.add_argument("--out", default="gpurun_out/sweep.json")
    ap.add_argument("--top", type=int, default=8)
    args = ap.parse_args()
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    P, B, N = args.problems, args.budgets, args.samples
    dev = torch.device("cuda:0")
    eng = Engine(device=0, timing=True)
    ans = torch.empty((P, B, N), dtype=torch.int32, device=dev)
    tok = torch.empty((P, B, N), dtype=torch.int32, device=dev) if args.tokens else None
    tr = torch.empty((P,), dtype=torch.int32, device=dev)
    counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
    cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
    lst = lambda s: [int(x) for x in s.split(",") if x]  # noqa: E731
    variants = []
    for c, t, w, u in itertools.product(lst(args.copies), lst(args.threads), lst(args.wg), lst(args.unroll)):
        lds = (1024 * c + 96) * 4
        if w > (160 * 1024) // lds or w > 2048 // t:
            continue                                     # would be clamped to an already-listed point
        for bal in lst(args.balance):
            for g in lst(args.grids):
                for sg in lst(args.stagger):
                    for pl in lst(args.plain):
                        variants.append((c, t, w, u, bal, g, sg, pl))
    results = []
    nbytes = P * B * N * 4 * (2 if args.tokens else 1)
    for d in lst(args.dists):
        eng.synth_fill_device(ans, tok, tr, P=P, B=B, N=N, seed=7, dist=d)
        eng.sync()
        times = {v: [] for v in variants}
        ref = None
        for r in range(args.rounds + 1):
            for v in variants:
                eng.set_tuning(*v[:2], v[2], v[3])
                eng.set_option("balance", v[4])
                eng.set_option("grid", v[5])
                eng.set_option("stagger_vecs", v[6])
                eng.set_option("plain_loads", v[7])
                counters.zero_()
                eng.aggregate_device(ans, tr, tokens=tok, counters=counters, cells=cells)
                eng.sync()
                ns, n = eng.drain_kernel_ns()
                if r:
                    times[v].append(ns / n)
                h = (int(counters.sum().item()), int(cells.to(torch.int64).sum().item()))
                if ref is None:
                    ref = h
                assert h == ref, f"variant {v} disagrees on dist {d}: {h} vs {ref}"
        for v in variants:
            med, mn = statistics.median(times[v]), min(times[v])
            results.append({"dist": d, "copies": v[0], "threads": v[1], "wg_per_cu": v[2], "unroll": v[3], "balance": v[4], "grid": v[5], "stagger": v[6], "plain": v[7],
                            "median_ms": med / 1e6, "min_ms": mn / 1e6, "GBps_median": nbytes / med, "GBps_best": nbytes / mn})
        best = sorted((r for r in results if r["dist"] == d), key=lambda r: r["median_ms"])[:args.top]
        print(f"dist {d}: top variants (copies, threads, wg/cu, unroll) -> GB/s median")
        for r in best:
            print(f"  R={r['copies']:2d} T={r['threads']:4d} wg={r['wg_per_cu']} U={r['unroll']} bal={r['balance']} grid={r['grid']} stag={r['stagger']} plain={r['plain']}  {r['GBps_median']:7.0f} GB/s  ({r['median_ms']:.3f} ms)")
        sys.stdout.flush()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"shape": [P, B, N], "bytes": nbytes, "tokens": args.tokens, "results": results}, f, indent=1)
    eng.close()


if __name__ == "__main__":
    main()
