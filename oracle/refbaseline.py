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

``--dropin`` (bench.py's ``cpu_baseline.dropin_loop``): after the reference has been timed on a cache, the SAME call on the SAME
cache is made once more with ``o1_dropin.install(o1, engine=Engine(timing=True))`` in force -- the product path: extract.py's key
scheme -> pinned vote tensors -> HOST-mode HIP engine -> host floats -- and its (accuracy, avg_tokens_used) must equal the
reference's.  The wall time is split into extract / engine call / kernel / floats (DropInConfig.timings).  ``--engine oracle``
puts the CPU oracle adapter behind the drop-in instead (no GPU: checks this script, measures only the extractor).

CLI: prints ONE JSON line.  Run by bench.py as a subprocess (the bench process holds a HIP runtime; the reference
starts ~300 threads per call).  Never imported by the product package.
"""
from __future__ import annotations

import json
import os
import time


_REBOUND = ("process_single_example", "run_experiments", "run_majority_vote_inference_experiments", "run_just_ask_nicely_experiments")


def same_accuracy(got: float, want: float, P: int = 30) -> bool:
    """Two accuracies as EXACT RATIONALS (as the tests compare them), not floats within a tolerance: the reference sums 1/len(modes)
    in thread-completion order (o1.py:236-239) and the drop-in in one canonical order, so the two floats may differ in the last ulps
    while naming the same rational k / (P * lcm of the tie sizes).  Each float is taken back to the nearest fraction with a denominator
    up to P * lcm(1..8) = 25 200 (ties of up to 8 answers; the floats are ~1e-15 from it, the next such fraction >= 7.8e-10 away) and
    the fractions must be EQUAL."""
    from fractions import Fraction
    D = P * 840
    fg, fw = Fraction(float(got)).limit_denominator(D), Fraction(float(want)).limit_denominator(D)
    # ... and each float must BE its fraction up to the rounding of a 30-term sum (1e-13 is ~1000 ulps; a float that is merely NEAR
    # a fraction with a small denominator is not that fraction)
    near = Fraction(1, 10 ** 13)
    return fg == fw and abs(Fraction(float(got)) - fg) <= near and abs(Fraction(float(want)) - fw) <= near


def _dropin_engine(kind: str):
    if kind == "oracle":
        from tests._adapters import OracleEngine
        return OracleEngine(), "oracle adapter (CPU; NOT the product: script self-check only)"
    from o1_inference_scaling_laws_amd.engine import Engine
    return Engine(timing=True), "HIP engine (libscvote.so, HOST mode)"


def _time_dropin(o1, engine, ds, cache, N, ref_acc, ref_avg, repeats):
    """o1.run_experiments(ds, cache, 2048, N) with the drop-in installed into the live reference module; then the module's own
    functions are put back."""
    from o1_inference_scaling_laws_amd import o1_dropin
    saved = {k: getattr(o1, k) for k in _REBOUND}
    try:
        cfg = o1_dropin.install(o1, engine=engine, batched=True)      # (cfg.save_cache = the no-op'd o1.save_cache)
        P = len(ds)
        best, split, got = None, None, None
        for _ in range(max(1, repeats)):
            for k in cfg.timings:
                cfg.timings[k] = 0
            c0 = time.perf_counter()
            got = o1.run_experiments(ds, cache, 2048, N)
            dt = time.perf_counter() - c0
            if best is None or dt < best:
                best, split = dt, dict(cfg.timings)
        acc, avg = got
        return {"seconds": best, "votes_per_s": P * N / best,
                "split_s": {"extract": split["extract"], "engine_call": split["engine"], "kernel": split["kernel"],
                            "staging_in_engine_call": max(0.0, split["engine"] - split["kernel"]), "floats": split["floats"],
                            "other": max(0.0, best - split["extract"] - split["engine"] - split["floats"])},
                "accuracy": acc, "avg_tokens_used": float(avg),
                "equal_to_reference": bool(same_accuracy(acc, ref_acc, P) and float(avg) == float(ref_avg)),
                "accuracy_compared_as": "exact rationals (Fraction.limit_denominator(P * lcm(1..8)))"}
    finally:
        for k, v in saved.items():
            setattr(o1, k, v)


def _family_records(o1, ds, cache):
    """The reference's two drivers as its import runs them (o1.py:314-315): 11 majority-vote budgets + 8 ask-nicely budgets = 19
    calls of run_experiments (or, with the batched drop-in, 3 engine calls).  The plot functions -- matplotlib, not the path -- are
    replaced by collectors; what they would have dumped (plot_helpers.py:59-60, 85-86: json.dump(results, f, indent=2)) is returned
    as the log text."""
    import contextlib
    import io
    got = {}
    saved = (o1.plot_majority_vote_graph, o1.plot_just_ask_nicely_graph)
    o1.plot_majority_vote_graph = lambda results, shade=False: got.__setitem__("majority_vote", list(results))
    o1.plot_just_ask_nicely_graph = lambda results, full=False: got.__setitem__("just_ask_nicely", list(results))
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            c0 = time.perf_counter()
            o1.run_majority_vote_inference_experiments(ds, cache)
            o1.run_just_ask_nicely_experiments(ds, cache)
            dt = time.perf_counter() - c0
    finally:
        o1.plot_majority_vote_graph, o1.plot_just_ask_nicely_graph = saved
    logs = {k: json.dumps([dict(r, avg_tokens_used=float(r["avg_tokens_used"])) for r in v], indent=2) for k, v in got.items()}
    return dt, got, logs


def _time_family(o1, engine, ds, cache, repeats):
    """The whole pipeline family end to end: the unmodified reference, then the same two driver calls with the drop-in installed
    (unbatched: the reference's own driver loops, 19 engine calls; batched: 3 engine calls)."""
    from o1_inference_scaling_laws_amd import o1_dropin
    ref_s, ref_rec, ref_logs = min((_family_records(o1, ds, cache) for _ in range(max(1, repeats))), key=lambda x: x[0])
    out = {"budgets": sum(len(v) for v in ref_rec.values()), "reference_seconds": ref_s}
    saved = {k: getattr(o1, k) for k in _REBOUND}
    for batched in (False, True):
        try:
            cfg = o1_dropin.install(o1, engine=engine, batched=batched)
            cfg.plot_majority_vote_graph = lambda results, shade=False: o1.plot_majority_vote_graph(results, shade)     # (the collectors installed
            cfg.plot_just_ask_nicely_graph = lambda results, full=False: o1.plot_just_ask_nicely_graph(results, full)   # by _family_records)
            _family_records(o1, ds, cache)                                   # warm
            best = None
            for _ in range(max(2, repeats)):
                for k in cfg.timings:
                    cfg.timings[k] = 0
                dt, rec, logs = _family_records(o1, ds, cache)
                if best is None or dt < best[0]:
                    best = (dt, rec, logs, dict(cfg.timings))
            dt, rec, logs, split = best
            close = all(same_accuracy(g["accuracy"], w["accuracy"], len(ds)) and float(g["avg_tokens_used"]) == float(w["avg_tokens_used"])
                        and g["token_limit"] == w["token_limit"] for k in ref_rec for g, w in zip(rec[k], ref_rec[k]))
            out["dropin_batched" if batched else "dropin_unbatched"] = {
                "seconds": dt, "speedup_vs_reference": ref_s / dt, "engine_calls": split["calls"],
                "split_s": {"extract": split["extract"], "engine_call": split["engine"], "kernel": split["kernel"], "floats": split["floats"]},
                "engine_call_us_mean": split["engine"] / max(split["calls"], 1) * 1e6,
                "records_equal_to_reference": bool(close and all(len(rec[k]) == len(ref_rec[k]) for k in ref_rec)),
                "log_bytes_equal_to_reference": bool(logs == ref_logs)}
        finally:
            for k, v in saved.items():
                setattr(o1, k, v)
    return out


def run(ns, seed: int, dist: int, repeats: int = 1, dropin: bool = False, engine_kind: str = "hip"):
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
    family = None
    engine, engine_what, dropin_error = None, None, None
    if dropin:
        try:
            engine, engine_what = _dropin_engine(engine_kind)
        except Exception as e:                                 # no library / no GPU: the reference timing is still reported
            dropin_error = f"{type(e).__name__}: {e}"
    with rh.imported_reference(ds, rh.build_cache(consts, ds, boot)) as (o1, _workdir):
        o1.save_cache = lambda cache, filename: None           # o1.py:242 -- file I/O, not the vote loop
        if engine is not None:                                 # first call of a process: context, staging pipeline, pinned slots
            cache8 = rh.build_cache(consts, ds, [(p, 2048, i, int(a0[p, 0, i]), int(t0[p, 0, i])) for p in range(P) for i in range(8)])
            _time_dropin(o1, engine, ds, cache8, 8, *o1.run_experiments(ds, cache8, 2048, 8), 1)
            # the reference's own shape: both driver families on the cache its import ran on (P = 30; N = 1 ... 8; 19 budgets)
            try:
                family = _time_family(o1, engine, ds, rh.build_cache(consts, ds, boot), repeats)
            except Exception as e:                             # reported, never fatal for the timing of the reference itself
                family = {"error": f"{type(e).__name__}: {e}"}
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
            row = {"P": P, "N": N, "seconds": best, "votes_per_s": P * N / best, "accuracy": acc,
                   "accuracy_matches_restatement": bool(ok)}
            if engine is not None:
                d = _time_dropin(o1, engine, ds, cache, N, acc, avg, max(2, repeats))
                d["speedup_vs_reference"] = best / d["seconds"]
                row["dropin"] = d
            results.append(row)
    if engine is not None and hasattr(engine, "close"):
        engine.close()
    return {"available": True, "reference": kind, "host_cores": os.cpu_count(), "results": results,
            "dropin_engine": engine_what, "dropin_error": dropin_error, "family": family,
            "what": "unmodified o1.run_experiments (o1.py:216-247) on a warm in-memory synthetic cache, save_cache no-op'd"}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, nargs="+", default=[256, 2048, 8192])
    ap.add_argument("--seed", type=int, default=20240914)
    ap.add_argument("--dist", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--dropin", action="store_true", help="also time the same calls with the drop-in installed (needs the GPU)")
    ap.add_argument("--engine", choices=["hip", "oracle"], default="hip", help="engine behind the drop-in (oracle: CPU self-check of this script)")
    args = ap.parse_args()
    print(json.dumps(run(args.N, args.seed, args.dist, args.repeats, args.dropin, args.engine)), flush=True)


if __name__ == "__main__":
    main()
