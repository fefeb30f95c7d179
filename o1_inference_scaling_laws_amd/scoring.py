"""Host-side float formulas shared by every path (GPU engine, tests' oracle adapter).

The device produces integers only; the reference's floats are computed HERE, once, in one canonical
order, so results do not depend on thread-completion order the way /root/reference/o1.py:236-239
does.
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np


def accuracy_from_tie_classes(tie_class_hits, num_problems: int) -> float:
    """o1.py:239,244: accuracy = sum(score) / len(dataset), score = 1/len(modes) on a hit.

    ``tie_class_hits[m]`` = number of problems whose truth is among m tied modes.  Summed in
    ascending m.  Equals the reference's float exactly whenever every tie size is a power of two
    (all partial sums are dyadic); otherwise it is one of the values the reference's
    nondeterministic accumulation order can produce, up to 1 ulp.
    """
    total_score = 0
    for m in range(1, len(tie_class_hits)):
        c = int(tie_class_hits[m])
        if c:
            total_score += c / m
    return total_score / num_problems


def exact_accuracy_from_tie_classes(tie_class_hits, num_problems: int) -> Fraction:
    total = Fraction(0)
    for m in range(1, len(tie_class_hits)):
        c = int(tie_class_hits[m])
        if c:
            total += Fraction(c, m)
    return total / num_problems


def avg_tokens_used(token_sum: int, num_problems: int) -> np.float64:
    """o1.py:245: np.mean([sum of tokens per problem]).  np.mean of an int list accumulates in
    float64; for totals below 2^53 that equals float(total) / P exactly."""
    return np.float64(int(token_sum)) / np.float64(num_problems)


def _pass_at_k_running(n, c, ks):
    """1 - prod_{i<k} (1 - c/(n-i)) for every k in ks (ascending), sharing one running log-sum.
    k terms per value, no cancellation: relative error ~ k * 1e-16."""
    c = np.asarray(c, dtype=np.int64)
    n = np.broadcast_to(np.asarray(n, dtype=np.int64), c.shape)
    nf, cf = n.astype(np.float64), c.astype(np.float64)
    ks = sorted(int(k) for k in ks)
    out, acc, i = {}, np.zeros(c.shape, dtype=np.float64), 0
    for k in ks:
        while i < k:
            den = nf - i
            ok = den > cf                                   # n - i > c: factor (n-c-i)/(n-i) is positive
            acc = acc + np.log1p(np.where(ok, -cf / np.where(ok, den, 1.0), 0.0))
            i += 1
        val = 1.0 - np.exp(acc)
        val = np.where((n - c) < k, 1.0, val)               # fewer than k wrong samples: certain hit
        val = np.where(c == 0, 0.0, val)
        out[k] = np.clip(val, 0.0, 1.0)
    return out


def pass_at_k(n, c, k: int):
    """Unbiased pass@k (Chen et al. 2021): 1 - C(n-c, k) / C(n, k), vectorised over n and c.

    NEW semantics (the reference has no pass@k).  n = votes in the cell, c = truth_count
    from the engine (integer, bit-exact).  The float is evaluated by THIS one function for every path
    (product form in log space, k terms).
    """
    return _pass_at_k_running(n, c, [k])[int(k)]


PASS_K_SWEEP = tuple(2 ** i for i in range(11))   # k = 1, 2, 4, ..., 1024 (BASELINE.json config 5)


def pass_at_k_sweep(n_valid, truth_count, ks=PASS_K_SWEEP):
    """truth_count [P, B] (from the cell table), n_valid [B] -> {k: mean-over-problems pass@k [B]}."""
    tc = np.asarray(truth_count, dtype=np.int64)
    n = np.asarray(n_valid, dtype=np.int64).reshape(1, -1)
    return {k: v.mean(axis=0) for k, v in _pass_at_k_running(n, tc, ks).items()}


def bootstrap_percentiles(counts, num_problems: int, lo=2.5, hi=97.5):
    """counts int64 [R, B, M] from scv_bootstrap -> (accuracy[R, B], lo[B], hi[B])."""
    counts = np.asarray(counts)
    R, B, _M = counts.shape
    acc = np.empty((R, B), dtype=np.float64)
    for r in range(R):
        for b in range(B):
            acc[r, b] = accuracy_from_tie_classes(counts[r, b], num_problems)
    return acc, np.percentile(acc, lo, axis=0), np.percentile(acc, hi, axis=0)


def bootstrap_percentiles_fast(counts, num_problems: int, lo=2.5, hi=97.5):
    """Same values as ``bootstrap_percentiles`` (adding c/m in ascending m; a zero count adds +0.0, which changes
    nothing), vectorised over the R x B resample table: 1000 resamples in well under a millisecond instead of
    ~20 ms of Python loops."""
    counts = np.asarray(counts)
    acc = np.zeros(counts.shape[:2], dtype=np.float64)
    for m in range(1, counts.shape[2]):
        acc = acc + counts[:, :, m].astype(np.float64) / m
    acc = acc / num_problems
    return acc, np.percentile(acc, lo, axis=0), np.percentile(acc, hi, axis=0)
