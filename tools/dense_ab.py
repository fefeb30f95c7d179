#!/usr/bin/env python3
"""4096 < N <= 8192: the register-streamed dense kernel with 8 parts per cell (option reg_n_max = 8192, the default) against the
streaming kernel (reg_n_max = 4096): parity vs the oracle, then timings on cold buffers.  (16 parts, N <= 16384, were measured too:
slower than the streaming kernel from 12288 up, and a count of 2^14 does not fit the kernel's key.)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine
    from oracle import coracle
    from tests._adapters import OracleEngine, assert_results_equal
    from regimes import run
    eng = Engine(device=0, timing=True)
    eng.set_option("reg_n_max", 8192)
    bad = 0
    for (P, B, N) in ((40, 3, 4097), (33, 2, 4608), (21, 3, 6001), (30, 2, 8192), (12, 3, 8189), (9, 2, 8191), (12, 3, 8195)):
        for dist in (0, 1, 2, 3, 5):
            a, t, tr = coracle.synth_fill(P, B, N, 5 + dist, dist, want_tokens=True)
            nv = np.array([N, N // 2 + 1, 7][:B], dtype=np.int32)
            for tok in (False, True):
                for nvv in (None, nv):
                    try:
                        assert_results_equal(eng.aggregate(a, tr, tokens=t if tok else None, n_valid=nvv),
                                             OracleEngine().aggregate(a, tr, tokens=t if tok else None, n_valid=nvv), check_tokens=tok)
                    except AssertionError as e:
                        bad += 1
                        print("MISMATCH", P, B, N, dist, tok, nvv, str(e)[:60])
    print("parity mismatches:", bad, flush=True)
    eng.close()
    rows = []
    for (P, B, N) in ((18000, 8, 4608), (18000, 8, 4501), (14000, 8, 6144), (16000, 4, 8192)):
        for tok in (False, True):
            rec = {"shape": [P, B, N], "tokens": tok}
            for label, nmax in (("stream", 4096), ("dense", 8192)):
                eng = Engine(device=0, timing=True)
                eng.set_option("reg_n_max", nmax)
                r = run(eng, torch, P // (2 if tok else 1), B, N, tok, dist=1, rounds=3)
                rec[label] = {"us": r["median_us"], "GBps": r["GBps"]}
                eng.close()
                torch.cuda.empty_cache()
            rows.append(rec)
            print(f"P={P} B={B} N={N} tok={int(tok)}  " + "  ".join(f"{k} {v['us']:8.1f} us {v['GBps']:7.1f} GB/s" for k, v in rec.items() if isinstance(v, dict)), flush=True)
    json.dump(rows, open("gpurun_out/dense_ab.json", "w"), indent=1)


if __name__ == "__main__":
    main()
