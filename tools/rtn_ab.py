#!/usr/bin/env python3
"""A/B on dense cells of 96 ... 1024 votes (VERDICT r4 next #3a): the register-resident cell kernels (scv_reg_cells / scv_reg_dense: ds_add without
return, pivots, a read-back pass for the counts) against RANKS FROM RETURNING LDS ATOMICS (scv_prefix_pool: one ds_add_rtn per vote, the returned
count is the vote's rank, max_count / n_modes / min_mode from running maxima of (rank, value) keys -- no read-back pass).  The second design exists
as the prefix-budget kernel; a dense tensor [P, B, N] is a pool [P * B, N] with the single budget N, so the same buffers go through both:
scv_aggregate_i32 on [P, B, N] and scv_aggregate_prefix_i32 on [P * B, N] with n_valid = [N] (16 and 32 lanes per cell).  Cold buffers
(> 256 MiB of distinct data per measurement), cell table written in both, counters compared."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine, counters_size
    eng = Engine(device=0, timing=True)
    dev = torch.device("cuda:0")
    out = []
    sizes = [int(x) for x in sys.argv[1:]] or [96, 128, 256, 512, 1024]
    for N in sizes:
        for dist in range(6):
            B = 4
            P = (410_000_000 // (4 * B * N)) // 64 * 64
            bufs = []
            for i in range(3):
                a = torch.empty((P, B, N), dtype=torch.int32, device=dev)
                tr = torch.empty((P,), dtype=torch.int32, device=dev)
                eng.synth_fill_device(a, None, tr, P=P, B=B, N=N, seed=11 + i, dist=dist)
                bufs.append((a, tr, tr.repeat_interleave(B).contiguous()))
            res = {"N": N, "dist": dist, "P": P, "B": B}
            ref = None
            for name, opts in (("reg_cells", None), ("rtn_g16", {"prefix_path": 4, "reg_shape": 16}), ("rtn_g32", {"prefix_path": 4, "reg_shape": 32})):
                if opts:
                    for k, v in opts.items():
                        eng.set_option(k, v)
                nb = 1 if opts else B
                counters = torch.zeros(counters_size(nb), dtype=torch.int64, device=dev)
                cells = torch.empty((P * B // nb, nb, 16), dtype=torch.uint8, device=dev)
                nvt = torch.tensor([N], dtype=torch.int32, device=dev)
                eng.sync(); eng.drain_kernel_ns()
                ts = []
                for r in range(4):
                    for (a, tr, tr_rep) in bufs:
                        counters.zero_()
                        if opts:
                            eng.aggregate_prefix_device(a.view(P * B, N), tr_rep, nvt, counters=counters, cells=cells)
                        else:
                            eng.aggregate_device(a, tr, counters=counters, cells=cells)
                        eng.sync()
                        ns, n = eng.drain_kernel_ns()
                        if r:
                            ts.append(ns)
                if opts:
                    for k in opts:
                        eng.set_option(k, 0)
                hits = counters.cpu().numpy()
                # hits per tie class summed over the budgets: the same number whichever way the cells are grouped
                tie = hits[: nb * 1025].reshape(nb, 1025).sum(axis=0)
                if ref is None:
                    ref = tie
                else:
                    assert (ref == tie).all(), (name, N, dist)
                med = statistics.median(ts)
                res[name + "_us"] = round(med / 1e3, 1)
                res[name + "_TBps"] = round(P * B * N * 4 / med / 1e3, 3)
            out.append(res)
            print(json.dumps(res), flush=True)
            del bufs
            torch.cuda.empty_cache()
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "rtn_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
