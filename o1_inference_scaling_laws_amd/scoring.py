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


def pass_at_k(n, c, k: int):
    """Unbiased pass@k (Chen et al. 2021): 1 - C(n-c, k) / C(n, k), vectorised over c.

    NEW semantics (the reference has no pass@k).  n = votes in the cell, c = truth_count
    from the engine.  Product form, evaluated in float64: prod_{i=n-c+1..n} (1 - k/i).
    """
    c = np.asarray(c, dtype=np.int64)
    n_arr = np.broadcast_to(np.asarray(n, dtype=np.int64), c.shape)
    out = np.ones(c.shape, dtype=np.float64)
    flat_c, flat_n, flat_o = c.reshape(-1), n_arr.reshape(-1), out.reshape(-1)
    for idx in range(flat_c.size):
        ci, ni = int(flat_c[idx]), int(flat_n[idx])
        if ni - ci < k:
            flat_o[idx] = 1.0
        elif ci == 0:
            flat_o[idx] = 0.0
        else:
            # log-space for large c: sum log1p(-k/i)
            i = np.arange(ni - ci + 1, ni + 1, dtype=np.float64)
            flat_o[idx] = 1.0 - float(np.exp(np.sum(np.log1p(-k / i))))
    return out


def bootstrap_percentiles(counts, num_problems: int, lo=2.5, hi=97.5):
    """counts int64 [R, B, M] from scv_bootstrap -> (accuracy[R, B], lo[B], hi[B])."""
    counts = np.asarray(counts)
    R, B, _M = counts.shape
    acc = np.empty((R, B), dtype=np.float64)
    for r in range(R):
        for b in range(B):
            acc[r, b] = accuracy_from_tie_classes(counts[r, b], num_problems)
    return acc, np.percentile(acc, lo, axis=0), np.percentile(acc, hi, axis=0)
