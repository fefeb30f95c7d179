"""TEST INFRASTRUCTURE -- the UNMODIFIED reference loop as a timed CPU baseline (bench.py's ``cpu_baseline`` leg).

What is timed is ``o1.run_experiments(dataset, cache, 2048, N)`` itself (/root/reference/o1.py:216-247): the nested
thread pools (o1.py:184-186, 232-234), the per-sample cache-key lookups (o1.py:85-88, 119), ``statistics.multimode``
(o1.py:202) and the completion-order float sum (o1.py:236-245) -- on a warm in-memory synthetic cache, with the
``save_cache`` JSON dump (o1.py:242: file I/O, out of the path's scope) rebound to a no-op.  SURVEY.md 8d calls this
"R0".  The module is imported by ``oracle/ref_harness.py`` from /root/reference where that exists and from the
bytecode in ``oracle/_ref`` (oracle/make_ref.py) on the GPU box.

The votes are the workload's own generator (include/scvote.h, distribution ``--dist``): problem p's N samples are
row (p, 0, :) of ``synth_fill(30, 1, N)``; the returned accuracy is compared with the rational value the restatement
(oracle/pyoracle.py) gives for the same votes, so the timing is of a run that produced the right answer.

CLI: prints ONE JSON line.  Run by bench.py as a subprocess (the bench process holds a HIP runtime; the reference
starts ~300 threads per call).  Never imported by the product package.
"""
from __future__ import annotations

import json
import os
import time


def run(ns, seed: int, dist: int, repeats: int = 1):
    from oracle import coracle, pyoracle
    from oracle import ref_harness as rh
    kind = rh.reference_kind()
    if kind is None:
        return {"available": False, "why": "neither /root/reference nor oracle/_ref (python -m oracle.make_ref) is present"}
    consts = rh.reference_constants()
    P = 30                                                    # o1.py:46 asserts 30 problems
    a0, t0, tr0 = coracle.synth_fill(P, 1, 8, seed, dist, want_tokens=True)
    ds = rh.make_dataset([str(int(t)) for t in tr0])
    # the import itself runs the reference's whole pipeline (o1.py:312-315): give it a small fully populated cache
    boot = [(p, T, 0, int(a0[p, 0, 0]), int(t0[p, 0, 0])) for p in range(P) for T in [2 ** i for i in range(4, 11)]]
    boot += [(p, 2048, i, int(a0[p, 0, i]), int(t0[p, 0, i])) for p in range(P) for i in range(8)]
    results = []
    with rh.imported_reference(ds, rh.build_cache(consts, ds, boot)) as (o1, _workdir):
        o1.save_cache = lambda cache, filename: None           # o1.py:242 -- file I/O, not the vote loop
        for N in ns:
            a, t, tr = coracle.synth_fill(P, 1, N, seed, dist, want_tokens=True)
            assert [int(x) for x in tr] == [int(x) for x in tr0]
            votes = [(a[p, 0].tolist(), t[p, 0].tolist()) for p in range(P)]
            cache = rh.build_cache(consts, ds, [(p, 2048, i, votes[p][0][i], votes[p][1][i]) for p in range(P) for i in range(N)])
            best = None
            for _ in range(max(1, repeats)):
                c0 = time.perf_counter()
                acc, avg = o1.run_experiments(ds, cache, 2048, N)
                dt = time.perf_counter() - c0
                best = dt if best is None else min(best, dt)
            want = pyoracle.exact_accuracy(votes, [int(x) for x in tr])
            want_avg = sum(sum(v[1]) for v in votes) / P
            ok = abs(acc - float(want)) < 1e-12 and float(avg) == want_avg
            results.append({"P": P, "N": N, "seconds": best, "votes_per_s": P * N / best, "accuracy": acc,
                            "accuracy_matches_restatement": bool(ok)})
    return {"available": True, "reference": kind, "host_cores": os.cpu_count(), "results": results,
            "what": "unmodified o1.run_experiments (o1.py:216-247) on a warm in-memory synthetic cache, save_cache no-op'd"}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, nargs="+", default=[256, 2048, 8192])
    ap.add_argument("--seed", type=int, default=20240914)
    ap.add_argument("--dist", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=1)
    args = ap.parse_args()
    print(json.dumps(run(args.N, args.seed, args.dist, args.repeats)), flush=True)


if __name__ == "__main__":
    main()
