#!/usr/bin/env python3
"""Prefix budgets over SHORT pools (the reference's own shape: N = 8 ... 4096 samples per problem, budgets 1, 2, 4 ... N):
scv_aggregate_prefix_i32 on pool[P, N] against the dense ragged expansion [P, B, N] + n_valid."""
import json
import os
import statistics as st
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    eng = Engine(device=0, timing=True)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    dev = torch.device("cuda:0")
    out = []
    for (P, N, with_tokens) in [(400000, 8, False), (400000, 16, False), (200000, 32, False), (200000, 64, False), (200000, 8, True), (200000, 16, True),
                                (200000, 32, True), (200000, 64, True), (200000, 128, False), (200000, 128, True), (100000, 256, False), (50000, 1024, False), (20000, 4096, False), (4000, 16384, False)]:
        nv = [1 << k for k in range(N.bit_length()) if (1 << k) <= N]
        B = len(nv)
        pool = torch.empty((P, 1, N), dtype=torch.int32, device=dev)
        tk = torch.empty((P, 1, N), dtype=torch.int32, device=dev) if with_tokens else None
        tr = torch.empty((P,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(pool, tk, tr, P=P, B=1, N=N, seed=4, dist=1)
        nvt = torch.tensor(nv, dtype=torch.int32, device=dev)
        counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
        cells = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
        res = {"shape": [P, B, N], "n_valid": "1,2,4..N", "tokens": with_tokens}
        for mode in ("prefix", "dense"):
            if mode == "dense":
                if P * B * N * 4 > 6e9:
                    continue
                dense = pool.expand(P, B, N).contiguous()
                dtk = tk.expand(P, B, N).contiguous() if with_tokens else None
            eng.sync(); eng.drain_kernel_ns()
            ts = []
            for r in range(6):
                counters.zero_()
                if mode == "prefix":
                    eng.aggregate_prefix_device(pool.view(P, N), tr, nvt, tokens=None if tk is None else tk.view(P, N), counters=counters, cells=cells)
                else:
                    eng.aggregate_device(dense, tr, tokens=dtk, n_valid=nvt, counters=counters, cells=cells)
                eng.sync()
                ns, n = eng.drain_kernel_ns()
                if r:
                    ts.append(ns / n)
            if mode == "prefix":
                ref = counters.clone()
            else:
                assert torch.equal(ref, counters), "prefix and dense counters differ"
                del dense, dtk
            res[mode + "_us"] = st.median(ts) / 1e3
        res["pool_GBps"] = P * N * 4 * (2 if with_tokens else 1) / (res["prefix_us"] * 1e3)
        res["votes_per_s"] = P * sum(nv) / (res["prefix_us"] * 1e-6)
        out.append(res)
        print(json.dumps(res), flush=True)
        del pool
        torch.cuda.empty_cache()
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prefix_small.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
