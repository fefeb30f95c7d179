"""The arithmetic of scv_prefix_pool (csrc/scvote_prefix.hip.h) restated in plain Python and checked against statistics.multimode -- the
reference's own call, /root/reference/o1.py:202 -- on CPU: ranks from returning atomics in ARBITRARY order between two boundaries, per-lane
running keys, the truth and the data pivot kept out of the histogram, and the merge of the three candidates at a boundary.  No GPU, no
library: this pins the algorithm the kernel implements (the kernel itself is compared bit for bit with the oracle by the -m gpu tests)."""
import random
import statistics

import pytest

RANK_SHIFT = 18


def prefix_stats_by_ranks(votes, truth, budgets, lanes=16, rng=None):
    """(max_count, n_modes, min_mode, truth_count, hit) of votes[:n] for every n in `budgets`, the way scv_prefix_pool gets them.

    * the PIVOT is the most frequent value among the first 16 votes that are not the truth (largest (count, -value) key; none if there is no
      such vote); votes equal to the truth or to the pivot never enter the histogram: their lanes count them;
    * every other vote goes into the histogram with a returning add: the returned old count + 1 is its RANK.  Between two boundaries the adds
      of the lanes land in an arbitrary order (here: shuffled) -- the statistics below do not depend on it;
    * lane l keeps K = max over its votes of (rank << 18 | 1023 - value) and S = (its largest rank, how many of its votes have it); a boundary
      takes the maximum of K over the lanes (M = its rank, the smallest value of that rank), the number of votes of rank M, the truth's and the
      pivot's counts, and merges the three candidates."""
    rng = rng or random.Random(0)
    head = [v for v in votes[:16] if v != truth]
    pivot = None
    if head:
        best = max((head[:i + 1].count(v), -v) for i, v in enumerate(head))            # rank in index order, then the smaller value
        pivot = -best[1]
    hist = {}
    K = [0] * lanes
    S = [(0, 0)] * lanes
    tcl = [0] * lanes
    pcl = [0] * lanes
    out = {}
    done = 0
    for n in sorted(set(min(max(b, 0), len(votes)) for b in budgets)):
        segment = list(range(done, n))
        rng.shuffle(segment)                                           # the order in which the lanes' atomics are served
        for i in segment:
            lane = (i // 4) % lanes                                    # vector k of lane l = votes (k G + l) 4 .. + 3
            v = votes[i]
            if v == truth:
                tcl[lane] += 1
            elif v == pivot:
                pcl[lane] += 1
            else:
                hist[v] = hist.get(v, 0) + 1
                r = hist[v]
                K[lane] = max(K[lane], (r << RANK_SHIFT) | (1023 - v))
                top, cnt = S[lane]
                S[lane] = (r, 1) if r > top else ((top, cnt + 1) if r == top else (top, cnt))
        done = n
        gK = max(K)
        Mh = gK >> RANK_SHIFT
        nmh = sum(c for (top, c) in S if top == Mh) if Mh else 0
        tc, pc = sum(tcl), sum(pcl)
        M = max(Mh, tc, pc)
        hit = 1 if (tc == M and M > 0) else 0
        pin = 1 if (pc == M and pc > 0) else 0
        n_modes = (nmh if Mh == M else 0) + hit + pin
        mm = 1023 - (gK & ((1 << RANK_SHIFT) - 1)) if (Mh == M and Mh) else 1024
        if hit:
            mm = min(mm, truth)
        if pin:
            mm = min(mm, pivot)
        out[n] = (M, n_modes, mm if M else 0xffff, tc, hit)
    return {b: out[min(max(b, 0), len(votes))] for b in budgets}


def reference_stats(votes, truth, n):
    pre = votes[:n]
    if not pre:
        return (0, 0, 0xffff, 0, 0)
    modes = statistics.multimode(pre)                                  # o1.py:202
    return (pre.count(modes[0]), len(modes), min(modes), pre.count(truth), 1 if truth in modes else 0)   # o1.py:204-213 as integers


CASES = []
_r = random.Random(20240914)
for _ in range(400):
    n = _r.choice([1, 2, 5, 16, 17, 31, 64, 100, 257, 700])
    dom = _r.choice([1, 2, 3, 7, 40, 1000])
    hot = _r.randrange(dom)
    votes = [hot if _r.random() < _r.choice([0.0, 0.3, 0.6, 1.0]) else _r.randrange(dom) for _ in range(n)]
    truth = _r.choice([hot, _r.randrange(dom), 1023, votes[0]])
    budgets = sorted({1, 2, 4, 8, 16, n, _r.randrange(0, n + 1), _r.randrange(0, n + 1), n + 3, 0})
    CASES.append((votes, truth, budgets, _r.choice([16, 32])))


@pytest.mark.parametrize("case", range(len(CASES)))
def test_ranks_pivot_and_truth_merge_equal_multimode(case):
    votes, truth, budgets, lanes = CASES[case]
    got = prefix_stats_by_ranks(votes, truth, budgets, lanes, random.Random(case))
    for b in budgets:
        assert got[b] == reference_stats(votes, truth, min(b, len(votes))), (b, votes, truth)


def test_the_order_of_the_atomics_does_not_matter():
    votes = [3, 3, 7, 3, 9, 7, 7, 3, 5, 5, 5, 5, 1, 7, 3, 9] * 9
    for seed in range(50):
        assert prefix_stats_by_ranks(votes, 5, [10, 50, 144], 16, random.Random(seed)) == prefix_stats_by_ranks(votes, 5, [10, 50, 144], 32, random.Random(1000 + seed))
