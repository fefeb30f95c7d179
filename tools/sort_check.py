#!/usr/bin/env python3
"""scv_sort_cells (one lane per cell, rows staged by LDS-DMA, sorted in registers) on the GPU box: a parity sweep against the
C oracle (every N = 4 ... 64 that is a multiple of 4, several B, ragged n_valid, tokens, every distribution, out-of-domain votes)
and an A/B timing against the kernels it replaces (option sort_n_max = 0).  One JSON file under gpurun_out/."""
from __future__ import annotations

import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parity(eng, big=False):
    from o1_inference_scaling_laws_amd.engine import AggregateResult
    from oracle import coracle
    from tests._adapters import OracleEngine, assert_results_equal
    bad = []
    n = 0
    rng = np.random.default_rng(3)
    for N in list(range(1, 65)) + []:
        for (P, B) in ((1, 1), (7, 3), (300, 4), (1000, 11), (5000, 8), (70000, 2)):
            if P * B * N > 6_000_000:
                continue
            for dist in (0, 1, 2, 3, 4, 5):
                if dist in (2, 5) and P > 300:
                    continue
                a, t, tr = coracle.synth_fill(P, B, N, 100 + N + dist, dist, want_tokens=True)
                cases = [(None, "full")]
                nv = rng.integers(0, N + 1, size=B).astype(np.int32)
                cases.append((nv, "ragged"))
                if B > 1:
                    nv2 = np.full(B, N, dtype=np.int32); nv2[0] = 1; nv2[-1] = max(1, N // 2)
                    cases.append((nv2, "ragged2"))
                for nvv, nm in cases:
                    for tok in (False, True):
                        n += 1
                        try:
                            got = eng.aggregate(a, tr, tokens=t if tok else None, n_valid=nvv)
                            want = OracleEngine().aggregate(a, tr, tokens=t if tok else None, n_valid=nvv)
                            assert_results_equal(got, want, check_tokens=tok)
                        except AssertionError as e:
                            bad.append((N, P, B, dist, nm, tok, str(e)[:80]))
    return n, bad


def main():
    import torch
    from o1_inference_scaling_laws_amd import _lib
    from o1_inference_scaling_laws_amd.engine import Engine
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from regimes import run
    out = {}
    if "--no-parity" not in sys.argv:
        eng = Engine(device=0, timing=True)
        eng.set_option("path", 5)
        n, bad = parity(eng)
        # device pointers that are not 16-byte aligned (views into larger buffers): the linear-image form, every alignment class
        from o1_inference_scaling_laws_amd.engine import AggregateResult, cells_from_torch
        from oracle import coracle
        from tests._adapters import OracleEngine, assert_results_equal
        dev = torch.device("cuda:0")
        for (P, B, N) in ((300, 3, 61), (300, 2, 64), (500, 3, 7), (200, 4, 32), (1000, 2, 16), (150, 3, 33), (90, 5, 48)):
            a, t, tr = coracle.synth_fill(P, B, N, 17 + N, 1, want_tokens=True)
            for off_a, off_t in ((1, 1), (1, 2), (3, 0), (0, 3), (2, 2)):
                ba = torch.zeros(P * B * N + 8, dtype=torch.int32, device=dev)
                bt = torch.zeros(P * B * N + 8, dtype=torch.int32, device=dev)
                va = ba[off_a:off_a + P * B * N].view(P, B, N); vt = bt[off_t:off_t + P * B * N].view(P, B, N)
                va.copy_(torch.from_numpy(a)); vt.copy_(torch.from_numpy(t))
                nv = np.array([N, max(0, N - 5), N // 3, 1, N][:B], dtype=np.int32)
                for tok in (False, True):
                    for nvv in (None, nv):
                        n += 1
                        counters, cells, ctok = eng.aggregate_device(va, torch.from_numpy(tr).to(dev), tokens=vt if tok else None,
                                                                     n_valid=None if nvv is None else torch.from_numpy(nvv).to(dev))
                        eng.sync()
                        got = AggregateResult.from_counters(counters.cpu().numpy(), P, B, cells_from_torch(cells), ctok.cpu().numpy() if tok else None)
                        want = OracleEngine().aggregate(a, tr, tokens=t if tok else None, n_valid=nvv)
                        try:
                            assert_results_equal(got, want, check_tokens=tok)
                        except AssertionError as e:
                            bad.append((N, P, B, "unaligned base", off_a, off_t, tok, str(e)[:80]))
        print(f"parity: {n} cases, {len(bad)} mismatches", flush=True)
        for b in bad[:20]:
            print("  MISMATCH", b, flush=True)
        print("sort_cells launches:", eng.stat("sort_cells"), flush=True)
        out["parity"] = {"cases": n, "mismatches": [list(map(str, b)) for b in bad[:50]]}
        # out-of-domain vote -> SCV_ERR_DOMAIN at sync
        a = np.zeros((100, 2, 32), dtype=np.int32); a[57, 1, 5] = 5000
        try:
            eng.aggregate(a, np.zeros(100, dtype=np.int32))
            print("domain: NO ERROR (wrong)")
            out["domain"] = "missing"
        except _lib.DomainError:
            print("domain: DomainError raised (ok)")
            out["domain"] = "ok"
        a[57, 1, 5] = 7
        eng.aggregate(a, np.zeros(100, dtype=np.int32), n_valid=np.array([32, 5], dtype=np.int32))    # beyond the prefix: no error
        eng.close()
    rows = []
    shapes = [(3200000, 4, 8), (1600000, 4, 16), (800000, 4, 32), (400000, 4, 48), (400000, 4, 64), (3200000, 4, 7), (800000, 4, 30), (800000, 4, 33), (400000, 4, 61), (400000, 4, 8), (400000, 4, 16), (400000, 4, 32), (200000, 4, 64)]
    for (P, B, N) in shapes:
        for tok in (False, True):
            if tok and P > 800000:
                continue
            for cells in (True, False):
                if tok and not cells:
                    continue
                rec = {"shape": [P, B, N], "tokens": tok, "cell_table": cells}
                for label, opts in (("lane/reg", {"sort_n_max": 0}), ("sort", {"sort_n_max": 64})):
                    eng = Engine(device=0, timing=True)
                    for k, v in opts.items():
                        eng.set_option(k, v)
                    r = run(eng, torch, P, B, N, tok, dist=1, rounds=4, want_cells=cells)
                    rec[label] = {"us": r["median_us"], "GBps": r["GBps"]}
                    eng.close()
                    torch.cuda.empty_cache()
                rows.append(rec)
                print(f"P={P:8d} B={B} N={N:3d} tok={int(tok)} cells={int(cells)}  " + "  ".join(f"{k} {v['us']:8.1f} us {v['GBps']:7.1f} GB/s" for k, v in rec.items() if isinstance(v, dict)), flush=True)
    out["ab"] = rows
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/sort_check.json", "w"), indent=1)


if __name__ == "__main__":
    main()
