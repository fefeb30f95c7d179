"""Test-only glue: present the CPU oracle behind the same duck-typed interface as the HIP engine,
and rebuild synthetic caches from the compact golden fixtures.  Never imported by the product."""
from __future__ import annotations

import numpy as np

from o1_inference_scaling_laws_amd.engine import CELL_DTYPE, AggregateResult
from o1_inference_scaling_laws_amd.extract import extraction_key, generation_key
from oracle import coracle

# any strings work: the key scheme is parametric in (model, prompt); the real constants live in the
# reference module and are only needed when talking to a real cache.
TEST_MODEL = "test-model"
TEST_PROMPT = "PROMPT<{token_limit}|{problem}|{token_limit}>"


class OracleEngine:
    """``.aggregate`` with the engine's signature, computed by oracle/scv_oracle.c."""

    def aggregate(self, answers, truth, tokens=None, n_valid=None, want_cells=True):
        out = coracle.aggregate(answers, truth, tokens=tokens, n_valid=n_valid)
        if out["rc"] != 0:
            raise ValueError(f"oracle rc={out['rc']}")
        P, B = out["cells"].shape
        cells = out["cells"].view(CELL_DTYPE) if want_cells else None
        return AggregateResult(P, B, cells, out["cell_tokens"] if tokens is not None else None,
                               out["tie_class_hits"], out["token_sum"], out["truth_count_sum"])


    def aggregate_prefix(self, pool, truth, n_valid, tokens=None, want_cells=True):
        """Oracle for the prefix mode = the dense oracle on answers[p, b, :] = pool[p, :]."""
        pool = np.asarray(pool, dtype=np.int32)
        B = len(n_valid)
        dense = np.ascontiguousarray(np.broadcast_to(pool[:, None, :], (pool.shape[0], B, pool.shape[1])))
        dtok = None if tokens is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(tokens, dtype=np.int32)[:, None, :], dense.shape))
        return self.aggregate(dense, truth, tokens=dtok, n_valid=np.asarray(n_valid, dtype=np.int32), want_cells=want_cells)


def make_dataset(truth_strings):
    return [{"problem": f"Problem {i}: compute f({i}).", "answer": s, "url": f"https://aops/2024_AIME_{i}"}
            for i, s in enumerate(truth_strings)]


def build_cache(dataset, samples, model=TEST_MODEL, prompt=TEST_PROMPT):
    """Same construction as oracle/ref_harness.build_cache, with test-local key constants."""
    cache = {}
    for (p, token_limit, idx, answer, tokens) in samples:
        if answer == "MISSING":
            continue
        content = f"[completion p={p} T={token_limit} i={idx}] final answer: {answer}"
        cache[generation_key(model, prompt, dataset[p]["problem"], token_limit, idx)] = {
            "content": content, "tokens": int(tokens)}
        cache[extraction_key(content)] = answer
    return cache


def raw_votes(case, p):
    """The (answers, tokens) lists the reference's loop o1.py:187-195 sees for problem p of a case."""
    N, T = case["N"], case["token_limit"]
    by_idx = {s[2]: s for s in case["samples"] if s[0] == p and s[1] == T}
    answers, tokens = [], []
    for idx in range(N):
        s = by_idx.get(idx)
        if s is None or s[3] is None or s[3] == "MISSING":
            answers.append(0)
            tokens.append(0)
        else:
            answers.append(s[3])
            tokens.append(s[4])
    return answers, tokens


def assert_results_equal(got: AggregateResult, want: AggregateResult, check_tokens=True):
    for f in ("max_count", "truth_count", "n_modes", "min_mode", "hit"):
        assert np.array_equal(got.cells[f], want.cells[f]), f
    assert np.array_equal(got.tie_class_hits, want.tie_class_hits)
    assert np.array_equal(got.truth_count_sum, want.truth_count_sum)
    if check_tokens:
        assert np.array_equal(got.token_sum, want.token_sum)
        if want.cell_tokens is not None:
            assert np.array_equal(got.cell_tokens, want.cell_tokens)


def assert_golden_case_via_prefix(engine, case, vt):
    """The reference's result for budget N of a golden case, obtained through the PREFIX entry point: the case's samples
    as one pool per problem, budgets {1, 2, N // 2, N} over it (the last one is the golden's).  ``engine`` is anything
    with ``aggregate_prefix`` (the HIP engine; the oracle adapter on CPU, which checks this helper itself)."""
    from fractions import Fraction
    pool = np.ascontiguousarray(vt.answers[:, 0, :])
    tpool = np.ascontiguousarray(vt.tokens[:, 0, :])
    N = pool.shape[1]
    nv = np.array(sorted({1, min(2, N), max(1, N // 2), N}), dtype=np.int32)
    res = engine.aggregate_prefix(pool, vt.truth, nv, tokens=tpool)
    b = len(nv) - 1                                                  # n_valid[b] == N
    for p, (num, den, tok) in enumerate(case["per_problem"]):
        cell = res.cells[p, b]
        assert (Fraction(1, int(cell["n_modes"])) if cell["hit"] else Fraction(0)) == Fraction(num, den), (case["name"], p)
        assert int(res.cell_tokens[p, b]) == tok, (case["name"], p)
    assert res.exact_accuracy(b) == Fraction(*case["accuracy_exact"]), case["name"]
    # budget 1 = the first sample alone: one mode, hit iff it is the truth (o1.py:202-206 on a single vote)
    first_is_truth = (pool[:, 0] == vt.truth)
    assert np.array_equal(res.cells["hit"][:, 0].astype(bool), first_is_truth) and (res.cells["n_modes"][:, 0] == 1).all()
