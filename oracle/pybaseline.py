"""TEST INFRASTRUCTURE -- the reference's own arithmetic as a timed CPU baseline (bench.py's cpu_baseline leg).

What is timed is oracle/pyoracle.process_votes: the line-by-line restatement of /root/reference/o1.py:181-213
that calls ``statistics.multimode`` exactly as the reference does (o1.py:202) on Python int lists, i.e. the
reference loop without its thread pools and cache-key lookups (SURVEY.md 8d "R1").  The unmodified reference
cannot travel to the GPU box (/root/reference does not exist there).  Never imported by the product package.
"""
from __future__ import annotations

import time


def _problem_cells(P, B, N, seed, dist, p_offset):
    """Votes of P problems as Python int lists (the reference holds Python ints): [(truth, [cell lists])]."""
    from oracle import coracle
    a, _, tr = coracle.synth_fill(P, B, N, seed, dist, p_offset=p_offset)
    return [(int(tr[p]), [a[p, b].tolist() for b in range(B)]) for p in range(P)]


def score_problem(truth, cells):
    """o1.py:181-213 per cell -> [(score, n_modes_if_hit)]; the token stream is not part of this baseline."""
    from oracle import pyoracle
    out = []
    for votes in cells:
        score, _ = pyoracle.process_votes(votes, (), truth)
        out.append(score)
    return out


def worker(args):
    """One pool task: build one problem's lists (untimed), run the reference arithmetic over its B cells (timed)."""
    B, N, seed, dist, p = args
    (truth, cells), = _problem_cells(1, B, N, seed, dist, p)
    t0 = time.perf_counter()
    scores = score_problem(truth, cells)
    return p, scores, time.perf_counter() - t0, B * N


def _noop(_):
    return 0


def single_core(B, N, seed, dist, p_offset, seconds, max_problems=64):
    """Run problems p_offset, p_offset+1, ... on THIS core until `seconds` of timed work; returns
    (votes_per_s, {p: scores}, problems, timed_seconds)."""
    results, votes, spent, p = {}, 0, 0.0, 0
    while p < max_problems and (p == 0 or spent < seconds):
        _, scores, dt, nv = worker((B, N, seed, dist, p_offset + p))
        results[p_offset + p] = scores
        votes += nv
        spent += dt
        p += 1
    return votes / spent, results, p, spent


def all_cores(B, N, seed, dist, p_offset, procs, problems_per_proc=1):
    """The same arithmetic with the problems spread over `procs` processes (multiprocessing, like the
    reference's per-problem parallelism o1.py:232-234 but on processes: its threads share the GIL).
    Rate = votes / wall time of the timed map (workers are started and warmed before the clock).
    Call only from a process that holds NO HIP runtime (this module's CLI): it forks."""
    import multiprocessing as mp
    tasks = [(B, N, seed, dist, p_offset + i) for i in range(procs * problems_per_proc)]
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_noop, range(procs * 4))  # start every worker
        t0 = time.perf_counter()
        out = pool.map(worker, tasks, chunksize=problems_per_proc)
        wall = time.perf_counter() - t0
    votes = sum(o[3] for o in out)
    busy = sum(o[2] for o in out)
    return votes / wall, {o[0]: o[1] for o in out}, wall, busy


def main():
    """CLI used by bench.py (a subprocess: the bench process holds a HIP runtime and must not fork workers):
    prints ONE JSON line with the single-core and all-cores rates and the per-problem scores."""
    import argparse
    import json
    import os
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, required=True)
    ap.add_argument("--N", type=int, required=True)
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--dist", type=int, required=True)
    ap.add_argument("--p-offset", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=5.0, help="single-core timed work")
    ap.add_argument("--procs", type=int, default=0, help="all-cores pool size (0 = skip)")
    args = ap.parse_args()
    rate1, res1, nprob, spent = single_core(args.B, args.N, args.seed, args.dist, args.p_offset, args.seconds)
    out = {"single": {"votes_per_s": rate1, "problems": nprob, "timed_s": spent},
           "scores": {str(p): [float(x) for x in sc] for p, sc in res1.items()}, "host_cores": os.cpu_count()}
    if args.procs > 0:
        rate, res, wall, busy = all_cores(args.B, args.N, args.seed, args.dist, args.p_offset, args.procs)
        out["all_cores"] = {"votes_per_s": rate, "procs": args.procs, "problems": len(res), "wall_s": wall, "busy_s": busy}
        out["scores"].update({str(p): [float(x) for x in sc] for p, sc in res.items()})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
