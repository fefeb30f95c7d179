#!/usr/bin/env python3
"""Generate tests/golden/golden_o1.json by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every case is a synthetic stand-in for helpers/response_cache.json (the real blob is missing from
the reference snapshot, /root/reference/.MISSING_LARGE_BLOBS:1) with the same schema
(o1.py:99-102, :119, :144).  For each case we record the compact inputs
(problem_idx, token_limit, idx, answer|None|"MISSING", tokens) and what the reference's own
functions returned:

  * o1.process_single_example(example, token_limit, cache, N)  -> per-problem (score, total_tokens)
  * o1.run_experiments(dataset, cache, token_limit, N)         -> (accuracy, avg_tokens_used)
  * the two helpers/results_log_*.json files the import-time pipeline wrote (bytes).

Scores are stored as exact unit fractions [num, den]; accuracy additionally as an exact rational,
because the reference's float accumulation order is nondeterministic.
"""
from __future__ import annotations

import json
import os
import random
import sys
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle import ref_harness as rh  # noqa: E402

P = 30  # o1.py:46


def peaked_answer(rng, truth, q, distractors):
    r = rng.random()
    if r < q:
        return truth
    if r < q + 0.3:
        return rng.choice(distractors)
    return rng.randrange(1000)


def pipeline_case(seed):
    """Inputs for the import-time run: maj-vote budgets 2^4..2^14 (o1.py:269) and ask-nicely
    2^4..2^11 (o1.py:297).  Re-seeded until every cell's tie size is a power of two so the
    accumulated accuracy float is order-independent and results_log bytes are reproducible."""
    while True:
        rng = random.Random(seed)
        truths = [rng.randrange(1000) for _ in range(P)]
        samples = []
        for p in range(P):
            q = rng.choice([0.15, 0.3, 0.5, 0.7, 0.9])
            dis = [rng.randrange(1000) for _ in range(2)]
            for T in [2 ** i for i in range(4, 11)]:
                samples.append((p, T, 0, peaked_answer(rng, truths[p], q * T / 2048 + 0.1, dis), rng.randrange(100, 3000)))
            for idx in range(8):
                samples.append((p, 2048, idx, peaked_answer(rng, truths[p], q, dis), rng.randrange(1500, 12000)))
        ok = True
        for p in range(P):
            pool = [s[3] for s in samples if s[0] == p and s[1] == 2048]
            for n in (1, 2, 4, 8):
                import statistics
                if len(statistics.multimode(pool[:n])) not in (1, 2, 4, 8):
                    ok = False
        if ok:
            return truths, samples
        seed += 1


def direct_cases():
    rng = random.Random(20240914)
    cases = []

    # --- ties, hand-built (N = 6 at the 2048 pool) ---
    truths = [rng.randrange(1, 1000) for _ in range(P)]
    rows = []
    for p in range(P):
        t = truths[p]
        a, b, c, d, e = [(t + k) % 1000 for k in (111, 222, 333, 444, 555)]
        pattern = [
            [t, t, a, a, b, c],      # 2-way tie incl. truth -> 1/2
            [t, t, a, a, b, b],      # 3-way tie incl. truth -> 1/3
            [a, a, b, b, c, c],      # 3-way tie excl. truth -> 0
            [t, t, t, a, b, c],      # clear correct majority -> 1
            [a, a, a, t, t, b],      # clear wrong majority   -> 0
            [t, a, b, c, d, e],      # 6-way tie incl. truth  -> 1/6
            [a, b, c, d, e, a],      # wrong plurality of 2   -> 0
            [t, a, t, a, t, a],      # interleaved 2-way tie  -> 1/2
            [a, t, b, t, c, t],      # truth wins with 3      -> 1
            [a, b, a, b, t, t],      # 3-way tie, truth last  -> 1/3
        ][p % 10]
        for i, v in enumerate(pattern):
            rows.append((p, 2048, i, v, 1000 + 37 * p + i))
    cases.append({"name": "ties_hand_built", "truths": [str(t) for t in truths], "samples": rows,
                  "token_limit": 2048, "N": 6})

    # --- failures: extraction None and missing generations become votes for 0 with 0 tokens ---
    truths = [0 if p % 3 == 0 else rng.randrange(1, 1000) for p in range(P)]
    rows = []
    for p in range(P):
        t = truths[p]
        for i in range(5):
            kind = (p + i) % 5
            if kind == 0:
                ans = None
            elif kind == 1:
                ans = "MISSING"
            elif kind == 2:
                ans = t
            elif kind == 3:
                ans = (t + 7) % 1000
            else:
                ans = 0
            rows.append((p, 2048, i, ans, 2000 + 11 * p + i))
    tstr = [("%03d" % t) if p % 4 == 1 else str(t) for p, t in enumerate(truths)]  # "033"-style strings
    cases.append({"name": "failures_vote_zero", "truths": tstr, "samples": rows, "token_limit": 2048, "N": 5})

    # --- out-of-domain ints compete as ordinary candidates (App. A2/A3) ---
    truths = [rng.randrange(1000) for _ in range(P)]
    rows = []
    ood = [-3, -1, 1000, 1001, 4096, 10 ** 12, -(10 ** 9), 123456789]
    for p in range(P):
        t = truths[p]
        o1_, o2_, o3_ = ood[p % len(ood)], ood[(p + 3) % len(ood)], ood[(p + 5) % len(ood)]
        r = (t + 13) % 1000
        pattern = [
            [o1_, o1_, o1_, t, t, r, o2_],       # out-of-domain value wins          -> 0
            [o1_, o1_, t, t, r, o2_, o3_],       # 2-way tie ood / truth             -> 1/2
            [o1_, o1_, o2_, o2_, t, t, r],       # 3-way tie two ood + truth         -> 1/3
            [t, t, t, o1_, o1_, o2_, o3_],       # truth wins over ood               -> 1
            [o1_, o2_, o3_, o1_, o2_, o3_, t],   # 3-way tie of ood only             -> 0
        ][p % 5]
        for i, v in enumerate(pattern):
            rows.append((p, 2048, i, v, 500 + i))
    cases.append({"name": "out_of_domain", "truths": [str(t) for t in truths], "samples": rows,
                  "token_limit": 2048, "N": 7})

    # --- random peaked data at several N (incl. non powers of two and N > 128) ---
    for N in (1, 2, 3, 5, 16, 64, 128, 200):
        truths = [rng.randrange(1000) for _ in range(P)]
        rows = []
        for p in range(P):
            q = rng.choice([0.05, 0.2, 0.4, 0.8])
            dis = [rng.randrange(1000) for _ in range(3)]
            for i in range(N):
                rows.append((p, 2048, i, peaked_answer(rng, truths[p], q, dis), rng.randrange(100, 12000)))
        cases.append({"name": f"random_peaked_N{N}", "truths": [str(t) for t in truths], "samples": rows,
                      "token_limit": 2048, "N": N})

    # --- N = 1 at a non-2048 token limit: key has no _idx suffix (o1.py:88) ---
    truths = [rng.randrange(1000) for _ in range(P)]
    rows = [(p, 64, 0, truths[p] if p % 2 else (truths[p] + 1) % 1000, 300 + p) for p in range(P)]
    cases.append({"name": "ask_nicely_T64_N1", "truths": [str(t) for t in truths], "samples": rows,
                  "token_limit": 64, "N": 1})

    # --- N larger than the pool: the extra indices are cache misses -> votes for 0 ---
    truths = [0 if p < 10 else rng.randrange(1, 1000) for p in range(P)]
    rows = []
    for p in range(P):
        for i in range(3):
            rows.append((p, 2048, i, truths[p] if i < 2 else 5, 777))
    cases.append({"name": "pool_shorter_than_N", "truths": [str(t) for t in truths], "samples": rows,
                  "token_limit": 2048, "N": 8})
    return cases


def generated_cases():
    """Cases whose samples come from a closed-form generator (tests/golden/gen.py): the fixture stores the
    generator parameters + what the reference returned, not the samples.  These reach the kernels the
    small cases cannot: N in the thousands (register-resident and streaming kernels), rows that are not
    16-byte aligned (N % 4 != 0), and problems with more than 24 distinct out-of-domain answers."""
    from tests.golden import gen
    specs = [
        ("many_out_of_domain_40", {"kind": "many_out_of_domain", "seed": 5, "P": 12, "N": 64, "distinct": 40}),
        ("many_out_of_domain_100", {"kind": "many_out_of_domain", "seed": 6, "P": 8, "N": 128, "distinct": 100}),
        ("large_n_3000", {"kind": "large_n", "seed": 9, "P": 6, "N": 3000}),
        ("large_n_4501", {"kind": "large_n", "seed": 10, "P": 6, "N": 4501}),
        ("large_n_17001", {"kind": "large_n", "seed": 11, "P": 6, "N": 17001}),
    ]
    cases = []
    for name, g in specs:
        params = {k: v for k, v in g.items() if k != "kind"}
        truths, samples = gen.GENERATORS[g["kind"]](**params)
        cases.append({"name": name, "gen": g, "truths": truths, "samples": samples, "token_limit": 2048, "N": g["N"]})
    return cases


def main():
    consts = rh.reference_constants()
    truths, samples = pipeline_case(7)
    dataset = rh.make_dataset([str(t) for t in truths])
    import_cache = rh.build_cache(consts, dataset, samples)
    out = {"generator": "tests/golden/make_golden.py", "reference": "hughbzhang/o1_inference_scaling_laws @ 2024-12-18",
           "python": sys.version.split()[0], "cases": []}
    with rh.imported_reference(dataset, import_cache) as (o1, workdir):
        logs = {}
        for name in ("results_log_majority_vote.json", "results_log_just_ask_nicely.json"):
            with open(os.path.join(workdir, "helpers", name)) as f:
                logs[name] = f.read()
        out["pipeline"] = {"truths": [str(t) for t in truths], "samples": samples, "results_logs": logs}

        for case in direct_cases() + generated_cases():
            ds = rh.make_dataset(case["truths"])
            cache = rh.build_cache(consts, ds, case["samples"])
            T, N = case["token_limit"], case["N"]
            per_problem = []
            for ex in ds:
                score, tokens = o1.process_single_example(ex, T, dict(cache), N)
                fr = Fraction(score).limit_denominator(4096)
                assert float(fr) == float(score)
                per_problem.append([fr.numerator, fr.denominator, int(tokens)])
            acc, avg = o1.run_experiments(ds, dict(cache), T, N)
            exact = sum(Fraction(a, b) for a, b, _ in per_problem) / len(ds)
            assert abs(float(exact) - acc) < 1e-12
            case["per_problem"] = per_problem
            case["accuracy_exact"] = [exact.numerator, exact.denominator]
            case["accuracy_live"] = repr(float(acc))
            case["avg_tokens_used"] = repr(float(avg))
            if "gen" in case:
                del case["samples"]                 # re-materialised from case["gen"] by tests/conftest.py
            out["cases"].append(case)
    path = os.path.join(HERE, "golden_o1.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(out['cases'])} direct cases")


if __name__ == "__main__":
    main()
