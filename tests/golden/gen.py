"""Closed-form sample generators for golden cases too large to store sample by sample.

Shared by tests/golden/make_golden.py (which feeds them to the UNMODIFIED reference) and by
tests/conftest.py (which re-materialises ``case["samples"]`` from ``case["gen"]``), so the fixture
holds only the generator parameters and what the reference returned.  Integer-only (splitmix64):
no dependence on the ``random`` module's algorithms.
"""
from __future__ import annotations

_M = (1 << 64) - 1
_G = 0x9E3779B97F4A7C15


def mix64(z: int) -> int:
    z &= _M
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _M
    z ^= z >> 31
    return z


def _u(seed, p, i, salt=0):
    return mix64(seed + _G * (((p * 1000003 + i) * 7 + salt) + 1))


def large_n(seed: int, P: int, N: int):
    """Long cells (N in the thousands): truths + samples at token_limit 2048.  Problem patterns p % 6:
    0 peaked on truth (q ~ 0.45) with distractors -> 1;   1 exact 2-way tie truth / other -> 1/2;
    2 exact 3-way tie without truth -> 0;   3 an out-of-domain value wins -> 0;
    4 uniform over 0..999 (whatever wins);   5 exact 4-way tie incl. truth + a failed-sample tail."""
    truths, samples = [], []
    for p in range(P):
        t = _u(seed, p, 0, 1) % 1000
        truths.append(t)
        a, b, c = (t + 111) % 1000, (t + 222) % 1000, (t + 333) % 1000
        kind = p % 6
        for i in range(N):
            u = _u(seed, p, i)
            tok = 100 + (u >> 40) % 11901
            if kind == 0:
                r = u % 100
                v = t if r < 45 else (a if r < 60 else (b if r < 70 else (u >> 8) % 1000))
            elif kind == 1:
                m = (N // 2) * 2
                v = (t, a)[i % 2] if i < m else b
            elif kind == 2:
                m = (N // 3) * 3
                v = (a, b, c)[i % 3] if i < m else t
            elif kind == 3:
                r = u % 100
                v = 10 ** 9 + 7 if r < 40 else (t if r < 70 else (-5 if r < 80 else (u >> 8) % 1000))
            elif kind == 4:
                v = (u >> 8) % 1000
            else:
                m = (N // 8) * 4                      # half the votes: 4-way tie; the rest spread thin
                v = (t, a, b, c)[i % 4] if i < m else 400 + (i % 301)
                if i >= N - 3:
                    v = None                          # failed samples -> votes for 0 with 0 tokens
            samples.append((p, 2048, i, v, tok))
    return [str(t) for t in truths], samples


def many_out_of_domain(seed: int, P: int, N: int, distinct: int):
    """More than 24 distinct out-of-domain answers per problem (the spare-bin dictionary overflows):
    `distinct` different values outside 0..999 (negative, >= 1000, huge) plus in-domain ones.
    Patterns p % 4: 0 truth wins; 1 one out-of-domain value wins; 2 tie truth / out-of-domain -> 1/2;
    3 three out-of-domain values tie -> 0."""
    truths, samples = [], []
    for p in range(P):
        t = _u(seed, p, 0, 2) % 1000
        truths.append(t)
        ood = []
        for k in range(distinct):
            u = _u(seed, p, k, 3)
            ood.append((1000 + u % 5000, -1 - (u % 900), 10 ** 12 + k, 1000 + k)[k % 4])
        ood = list(dict.fromkeys(ood))
        kind = p % 4
        votes = list(ood)                              # every distinct value appears at least once
        if kind == 0:
            votes += [t] * 3
        elif kind == 1:
            votes += [ood[5]] * 2 + [t]
        elif kind == 2:
            votes += [ood[7], t, t]
        else:
            votes += [ood[1], ood[2], ood[3], t]
        while len(votes) < N:
            votes.append((_u(seed, p, len(votes), 4) >> 8) % 1000 if len(votes) % 2 else ood[len(votes) % len(ood)] + 10 ** 6 * len(votes))
        votes = votes[:N]
        # shuffle deterministically (multimode is order-independent; the cache-key idx is not)
        order = sorted(range(N), key=lambda i: _u(seed, p, i, 5))
        for i, src in enumerate(order):
            samples.append((p, 2048, i, votes[src], 200 + (i * 37) % 1000))
    return [str(t) for t in truths], samples


GENERATORS = {"large_n": large_n, "many_out_of_domain": many_out_of_domain}


def materialise(case: dict) -> dict:
    """Fill case['truths'] / case['samples'] from case['gen'] = {'kind': ..., **params} (in place)."""
    g = case.get("gen")
    if g and "samples" not in case:
        params = {k: v for k, v in g.items() if k != "kind"}
        truths, samples = GENERATORS[g["kind"]](**params)
        assert truths == case["truths"], "generator drifted from the committed fixture"
        case["samples"] = samples
    return case
