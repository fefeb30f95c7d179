"""TEST INFRASTRUCTURE -- pure-Python restatement of the reference's aggregation arithmetic.

Follows /root/reference/o1.py line by line, with the thread pools and the cache lookups removed
(they only produce the (answer, tokens) pairs) and with ``statistics.multimode``
called exactly as the reference calls it (o1.py:202).  Accepts arbitrary Python ints, so it is
also the oracle for the out-of-domain dictionary encoding done by the extractor.

Never imported by the product package.
"""
from __future__ import annotations

import statistics
from fractions import Fraction

import numpy as np


def process_votes(answers, tokens, truth):
    """o1.py:181-213 on already-resolved (answer, tokens) pairs.

    Returns (score, total_tokens) with the reference's types: score is the int 0 or the float
    1/len(modes) (o1.py:204,210); total_tokens is a Python int (o1.py:182,195).
    """
    answers = list(answers)                       # o1.py:181,194
    total_tokens = 0                              # o1.py:182
    for t in tokens:
        total_tokens += t                         # o1.py:195
    majority_answers = statistics.multimode(answers)   # o1.py:202
    score = 0                                     # o1.py:204
    if int(truth) in majority_answers:            # o1.py:206
        score = 1 / len(majority_answers)         # o1.py:210
    return score, total_tokens                    # o1.py:213


def cell_integers(answers, truth):
    """The integer cell record of include/scvote.h from multimode (arbitrary ints allowed)."""
    answers = list(answers)
    modes = statistics.multimode(answers)
    max_count = answers.count(modes[0]) if modes else 0
    return {
        "max_count": max_count,
        "truth_count": answers.count(int(truth)),
        "n_modes": len(modes),
        "hit": int(int(truth) in modes),
        "modes": modes,
    }


def run_experiments_votes(per_problem, truths, order=None):
    """o1.py:229-247 on resolved votes.  per_problem[p] = (answers, tokens).

    ``order`` is the completion order of the problem futures (o1.py:236 as_completed is
    nondeterministic).  Default: problem order.
    """
    P = len(per_problem)
    total_score = 0                               # o1.py:229
    actual_tokens_used = []                       # o1.py:230
    for p in (order if order is not None else range(P)):
        score, tokens = process_votes(per_problem[p][0], per_problem[p][1], truths[p])
        if score > 0:                             # o1.py:238
            total_score += score                  # o1.py:239
        actual_tokens_used.append(tokens)         # o1.py:240
    accuracy = total_score / P                    # o1.py:244
    avg_tokens_used = np.mean(actual_tokens_used)  # o1.py:245
    return accuracy, avg_tokens_used


def exact_accuracy(per_problem, truths):
    """Order-independent rational form of o1.py:239,244 (for comparisons that must not depend on
    float summation order)."""
    total = Fraction(0)
    for (answers, _tok), t in zip(per_problem, truths):
        modes = statistics.multimode(list(answers))
        if int(t) in modes:
            total += Fraction(1, len(modes))
    return total / len(per_problem)


def majority_vote_budgets(shade_regions=False):
    """o1.py:266-276: [(token_limit, actual_token_limit, N)]."""
    token_limits = [2 ** i for i in range(4, 19)] if shade_regions else [2 ** i for i in range(4, 15)]
    out = []
    for token_limit in token_limits:
        actual_token_limit = min(2 ** 11, token_limit)   # o1.py:274
        N = token_limit // actual_token_limit            # o1.py:276
        out.append((token_limit, actual_token_limit, N))
    return out


def just_ask_nicely_budgets(run_full_range=False):
    """o1.py:297-302: [(token_limit, token_limit, 1)]."""
    token_limits = [2 ** i for i in range(4, 12)]
    if run_full_range:
        token_limits = [2 ** i for i in range(20)]
    return [(t, t, 1) for t in token_limits]
