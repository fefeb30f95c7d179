#!/usr/bin/env python3
"""A/B of the register-resident kernels' second pivot (option "reg_pivots": 0 = two pivots per lane, 1 = the second one off) over the distributions:
one table row per (N, dist), microseconds per launch.  Evidence for DESIGN.md section 3.2 (profiles/r03_pivots_ab.log)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine
    from regimes import run
    eng = Engine(device=0, timing=True)
    names = {0: "D0 uniform", 1: "D1 peaked", 2: "D2 degenerate", 3: "D3 tie", 4: "D4 wrong majority", 5: "D5 degenerate-wrong"}
    print(f"{'shape':24s} {'dist':20s} " + " ".join(f"{'pivots=' + str(k):>12s}" for k in (0, 1)) + "   (us per launch; GB/s with both pivots)")
    for (P, B, N) in ((200000, 4, 64), (100000, 4, 256), (50000, 4, 1024), (40000, 4, 2048), (20000, 8, 4096)):
        for d in (1, 0, 2, 3, 4, 5):
            row = []
            gbs = 0
            for k in (0, 1):
                eng.set_option("reg_pivots", k)
                r = run(eng, torch, P, B, N, False, dist=d, rounds=3)
                row.append(r["median_us"])
                if k == 0:
                    gbs = r["GBps"]
            eng.set_option("reg_pivots", 0)
            print(f"{str([P, B, N]):24s} {names[d]:20s} " + " ".join(f"{x:12.1f}" for x in row) + f"   {gbs:8.0f} GB/s", flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
