#!/usr/bin/env python3
"""Throughput of the hot path outside the headline regime: tokens stream, few-cells (C2), small N.
Cold-cache discipline: cycles enough distinct buffers to exceed the 256 MiB Infinity Cache."""
from __future__ import annotations

import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(eng, torch, P, B, N, tokens, dist=1, nbuf=None, rounds=6, n_valid=None, want_cells=True):
    from o1_inference_scaling_laws_amd.engine import counters_size
    dev = torch.device("cuda:0")
    per = P * B * N * 4 * (2 if tokens else 1)
    if nbuf is None:
        nbuf = max(1, min(8, int(600e6 // max(per, 1)) + 1))
    bufs = []
    for i in range(nbuf):
        a = torch.empty((P, B, N), dtype=torch.int32, device=dev)
        t = torch.empty((P, B, N), dtype=torch.int32, device=dev) if tokens else None
        tr = torch.empty((P,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(a, t, tr, P=P, B=B, N=N, seed=11 + i, dist=dist)
        bufs.append((a, t, tr))
    nv = None if n_valid is None else torch.tensor(n_valid, dtype=torch.int32, device=dev)
    counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
    cells = torch.empty((P, B, 4 if getattr(eng, "packed_cells", False) else 16), dtype=torch.uint8, device=dev) if want_cells else False   # False: counters only, no cell table
    eng.sync(); eng.drain_kernel_ns()
    times = []
    for r in range(rounds + 1):
        for (a, t, tr) in bufs:
            counters.zero_()
            eng.aggregate_device(a, tr, tokens=t, n_valid=nv, counters=counters, cells=cells)
            eng.sync()
            ns, n = eng.drain_kernel_ns()
            if r:
                times.append(ns / n)
    med = statistics.median(times)
    votes = P * B * N if n_valid is None else P * sum(min(v, N) for v in n_valid)
    return {"shape": [P, B, N], "tokens": tokens, "dist": dist, "cell_table": bool(want_cells), "buffers": nbuf, "median_us": med / 1e3, "min_us": min(times) / 1e3,
            "GBps": votes * 4 * (2 if tokens else 1) / med, "votes_per_s": votes / (med * 1e-9)}


def main():
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine
    eng = Engine(device=0, timing=True)
    out = []
    only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]      # substrings of case names
    cases = [
        ("headline slab", 256, 8, 1 << 20, False, None),
        ("tokens stream", 128, 8, 1 << 20, True, None),
        ("C2 30x8x2^17", 30, 8, 1 << 17, False, None),
        ("C2 + tokens", 30, 8, 1 << 17, True, None),
        ("P=30 B=1 N=2^20", 30, 1, 1 << 20, False, None),
        ("P=1 B=1 N=2^24", 1, 1, 1 << 24, False, None),
        ("ragged D4 N>>(7-b)", 256, 8, 1 << 20, False, [(1 << 20) >> (7 - b) for b in range(8)]),
        ("mid N=16384 P=8000", 8000, 4, 16384, False, None),
        ("mid N=8192 P=16000", 16000, 4, 8192, False, None),
        ("mid N=4096 P=20000", 20000, 8, 4096, False, None),
        ("mid N=4096 + tokens", 20000, 4, 4096, True, None),
        ("mid N=2048 P=40000", 40000, 4, 2048, False, None),
        ("mid N=1024 P=50000", 50000, 4, 1024, False, None),
        ("mid N=1024 D2 degenerate", 50000, 4, 1024, False, None, 2),
        ("mid N=1000 (rows not 16-B aligned: N=1001)", 50000, 4, 1001, False, None),
        ("small N=512 P=100000", 100000, 4, 512, False, None),
        ("small N=256 P=100000", 100000, 4, 256, False, None),
        ("small N=256 + tokens", 100000, 4, 256, True, None),
        ("small N=128 P=200000", 200000, 4, 128, False, None),
        ("small N=64 P=200000", 200000, 4, 64, False, None),
        ("tiny N=16 P=400000", 400000, 4, 16, False, None),
        ("tiny N=8 P=400000", 400000, 4, 8, False, None),
        ("reference family 30x11x8", 30, 11, 8, True, [1] * 8 + [2, 4, 8]),
        # rows that are not 16-byte aligned (the reference's N is arbitrary, o1.py:276): aligned supersets, not dword loads
        ("N=1000 (aligned, cells not full)", 50000, 4, 1000, False, None),
        ("N=1001 (unaligned rows)", 50000, 4, 1001, False, None),
        ("N=1001 + tokens (unaligned)", 50000, 4, 1001, True, None),
        ("N=253 (unaligned rows)", 100000, 4, 253, False, None),
        ("N=4093 (unaligned rows)", 20000, 8, 4093, False, None),
        ("N=4501 (streaming, unaligned)", 18000, 8, 4501, False, None),
        ("N=4608 (streaming, aligned)", 18000, 8, 4608, False, None),
        # counters only (no cell table written): what the drop-in's run_experiments needs
        ("tiny N=8 counters only", 400000, 4, 8, False, None, 1, False),
        ("tiny N=16 counters only", 400000, 4, 16, False, None, 1, False),
        ("tiny N=32 counters only", 400000, 4, 32, False, None, 1, False),
        ("small N=64 counters only", 200000, 4, 64, False, None, 1, False),
        ("small N=256 counters only", 100000, 4, 256, False, None, 1, False),
        # the reference's own range at sizes where the launch is not the measurement (200-400 MB of votes; the rows above are 50 MB)
        ("tiny N=8 P=3.2M", 3200000, 4, 8, False, None),
        ("tiny N=8 P=3.2M counters only", 3200000, 4, 8, False, None, 1, False),
        ("tiny N=16 P=1.6M", 1600000, 4, 16, False, None),
        ("tiny N=16 P=1.6M counters only", 1600000, 4, 16, False, None, 1, False),
        ("tiny N=32 P=800k", 800000, 4, 32, False, None),
        ("tiny N=32 P=800k counters only", 800000, 4, 32, False, None, 1, False),
        ("tiny N=32 P=800k + tokens", 800000, 4, 32, True, None),
        ("small N=48 P=400k", 400000, 4, 48, False, None),
        ("small N=64 P=400k", 400000, 4, 64, False, None),
        ("small N=64 P=400k counters only", 400000, 4, 64, False, None, 1, False),
        ("small N=64 P=400k + tokens", 400000, 4, 64, True, None),
        ("N=7 P=3.2M (unaligned rows)", 3200000, 4, 7, False, None),
        ("N=30 P=800k (unaligned rows)", 800000, 4, 30, False, None),
        ("N=61 P=400k (unaligned rows)", 400000, 4, 61, False, None),
        ("N=64 D3 tie", 400000, 4, 64, False, None, 3),
        ("N=64 D5 degenerate-wrong", 400000, 4, 64, False, None, 5),
        ("N=96 P=200k", 200000, 4, 96, False, None),
        # round 4: the 48-vote shape of the sorted-cells kernel (33 ... 48 votes used to sort 64 slots)
        ("small N=36 P=500k", 500000, 4, 36, False, None),
        ("small N=40 P=500k", 500000, 4, 40, False, None),
        ("small N=44 P=400k", 400000, 4, 44, False, None),
        ("small N=48 P=400k counters only", 400000, 4, 48, False, None, 1, False),
        ("small N=48 P=400k + tokens", 400000, 4, 48, True, None),
        ("N=45 P=400k (unaligned rows)", 400000, 4, 45, False, None),
        ("N=48 D0 uniform", 400000, 4, 48, False, None, 0),
        ("N=48 D3 tie", 400000, 4, 48, False, None, 3),
        ("N=48 D5 degenerate-wrong", 400000, 4, 48, False, None, 5),
        # round 4: the reference's MOST COMMON cell sizes -- N = 1 for all 8-20 ask-nicely budgets (o1.py:302), N = 1 x 8, 2, 4 (o1.py:276)
        ("N=1 P=12.8M", 12800000, 8, 1, False, None),
        ("N=1 P=12.8M counters only", 12800000, 8, 1, False, None, 1, False),
        ("N=1 P=12.8M + tokens", 12800000, 8, 1, True, None),
        ("N=1 P=12.8M + tokens counters only", 12800000, 8, 1, True, None, 1, False),
        ("N=2 P=12.8M", 12800000, 4, 2, False, None),
        ("N=2 P=12.8M counters only", 12800000, 4, 2, False, None, 1, False),
        ("N=2 P=12.8M + tokens", 12800000, 4, 2, True, None),
        ("N=4 P=6.4M", 6400000, 4, 4, False, None),
        ("N=4 P=6.4M counters only", 6400000, 4, 4, False, None, 1, False),
        ("N=4 P=6.4M + tokens", 6400000, 4, 4, True, None),
        ("N=4 P=6.4M + tokens counters only", 6400000, 4, 4, True, None, 1, False),
    ]
    # the hot value is NOT the truth (D3 exact ties, D4 a confidently wrong majority, D5 all votes one wrong value) and D0
    for (P, B, N) in ((100000, 4, 256), (50000, 4, 1024), (20000, 8, 4096)):
        for d, nm in ((0, "D0 uniform"), (3, "D3 tie"), (4, "D4 wrong majority"), (5, "D5 degenerate-wrong")):
            cases.append((f"N={N} {nm}", P, B, N, False, None, d))
    for case in cases:
        name, P, B, N, tok, nv = case[:6]
        if only and not any(o in name for o in only):
            continue
        r = run(eng, torch, P, B, N, tok, n_valid=nv, dist=case[6] if len(case) > 6 else 1, want_cells=case[7] if len(case) > 7 else True)
        r["name"] = name
        out.append(r)
        print(f"{name:28s} {str(r['shape']):22s} tok={int(tok)}  {r['median_us']:10.1f} us  {r['GBps']:8.1f} GB/s  {r['votes_per_s']:.3e} votes/s", flush=True)
        torch.cuda.empty_cache()
    if only:
        return
    # prefix budgets over one pool: one pass vs the dense expansion (reported separately: different bytes)
    from o1_inference_scaling_laws_amd.engine import counters_size
    import statistics as st
    dev = torch.device("cuda:0")
    for (P, N, nv) in [(2048, 1 << 20, [(1 << 20) >> (7 - b) for b in range(8)]), (2048, 1 << 20, [((b + 1) << 20) // 8 for b in range(8)])]:
        pool = torch.empty((P, 1, N), dtype=torch.int32, device=dev)
        tr = torch.empty((P,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(pool, None, tr, P=P, B=1, N=N, seed=4, dist=1)
        nvt = torch.tensor(nv, dtype=torch.int32, device=dev)
        counters = torch.zeros(counters_size(len(nv)), dtype=torch.int64, device=dev)
        cells = torch.empty((P, len(nv), 16), dtype=torch.uint8, device=dev)
        eng.sync(); eng.drain_kernel_ns()
        ts = []
        for r in range(6):
            counters.zero_()
            eng.aggregate_prefix_device(pool.view(P, N), tr, nvt, counters=counters, cells=cells)
            eng.sync()
            ns, n = eng.drain_kernel_ns()
            if r:
                ts.append(ns / n)
        med = st.median(ts)
        votes = P * sum(nv)
        r = {"name": f"prefix one-pass n_valid={'pow2' if nv[-1] >= 2 * nv[-2] else 'linear'}", "shape": [P, len(nv), N], "median_us": med / 1e3,
             "GBps": P * max(nv) * 4 / med, "votes_per_s": votes / (med * 1e-9), "dense_equivalent_GBps": votes * 4 / med}
        out.append(r)
        print(f"{r['name']:28s} {str(r['shape']):22s}        {r['median_us']:10.1f} us  {r['GBps']:8.1f} GB/s of pool bytes  {r['votes_per_s']:.3e} votes/s (dense-equivalent {r['dense_equivalent_GBps']:.0f} GB/s)", flush=True)
        del pool
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/regimes.json", "w"), indent=1)


if __name__ == "__main__":
    main()
