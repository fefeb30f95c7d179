#!/usr/bin/env python3
"""Run ONE (P, B, N) shape of the hot path a few times on cold buffers -- the unit rocprofv3 wraps for
per-regime PMC passes (tools/prof_regimes.sh).  Prints one JSON line with the hipEvent timing."""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, required=True)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--N", type=int, required=True)
    ap.add_argument("--tokens", action="store_true")
    ap.add_argument("--dist", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (scv_set_option)")
    args = ap.parse_args()
    import torch
    from o1_inference_scaling_laws_amd.engine import Engine
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from regimes import run
    eng = Engine(device=0, timing=True)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    r = run(eng, torch, args.P, args.B, args.N, args.tokens, dist=args.dist, rounds=args.rounds)
    r["opts"] = args.opt
    print(json.dumps(r))


if __name__ == "__main__":
    main()
