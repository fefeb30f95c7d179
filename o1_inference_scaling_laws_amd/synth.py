"""CPU mirror (numpy, uint64) of the on-device synthetic generator, spec in include/scvote.h
(scv_synth_fill_i32).  Bit-identical to the HIP kernel; used to regenerate sub-samples of tensors
that only ever exist in HBM."""
from __future__ import annotations

import numpy as np

G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_KP = np.uint64(0x5851F42D4C957F2D)
_MASK32 = np.uint64(0xFFFFFFFF)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z ^ (z >> np.uint64(30))
        z = z * _M1
        z = z ^ (z >> np.uint64(27))
        z = z * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _mulhi32(a, n):
    return ((np.asarray(a, dtype=np.uint64) & _MASK32) * np.uint64(n)) >> np.uint64(32)


def problem_params(seed: int, p):
    """truth, q_num, distractors[4] for global problem indices p (array)."""
    p = np.asarray(p, dtype=np.uint64)
    with np.errstate(over="ignore"):
        k = mix64((np.uint64(seed) ^ _KP) + G * (p + np.uint64(1)))
        truth = _mulhi32(k, 1000)
        q_num = np.uint64(1) + (k >> np.uint64(32)) % np.uint64(7)
        d = np.stack([_mulhi32(mix64(k + G * np.uint64(j + 1)), 1000) for j in range(4)], axis=-1)
    return truth.astype(np.int64), q_num.astype(np.int64), d.astype(np.int64)


def truth(P: int, seed: int, p_offset: int = 0):
    return problem_params(seed, np.arange(p_offset, p_offset + P))[0].astype(np.int32)


def fill(P: int, B: int, N: int, seed: int, dist: int, p_offset: int = 0, want_tokens: bool = False):
    """answers int32[P,B,N], tokens int32[P,B,N] | None, truth int32[P]."""
    ps = np.arange(p_offset, p_offset + P, dtype=np.uint64)
    tr, q_num, d = problem_params(seed, ps)
    answers = np.empty((P, B, N), dtype=np.int32)
    tokens = np.empty((P, B, N), dtype=np.int32) if want_tokens else None
    i = np.arange(N, dtype=np.uint64)
    T5 = np.uint64(214748364)
    with np.errstate(over="ignore"):
        for pl in range(P):
            p = ps[pl]
            for b in range(B):
                e0 = (p * np.uint64(B) + np.uint64(b)) * np.uint64(N)
                u = mix64(np.uint64(seed) + G * (e0 + i + np.uint64(1)))
                hi = u >> np.uint64(32)
                uv = _mulhi32(u, 1000)
                if dist == 0:
                    v = uv
                elif dist == 1:
                    t0 = np.uint64((int(q_num[pl]) * 429496729) & 0xFFFFFFFF)
                    x = (hi - t0) & _MASK32
                    j = np.minimum(x // T5, np.uint64(3)).astype(np.int64)
                    v = np.where(hi < t0, np.uint64(tr[pl]), np.where(x < np.uint64(4) * T5, d[pl][j].astype(np.uint64), uv))
                elif dist == 2:
                    v = np.full(N, tr[pl], dtype=np.uint64)
                elif dist == 3:
                    m = 2 + (int(p) & 1)
                    base = (int(tr[pl]) + 500) % 1000 if (int(p) >> 1) & 1 else int(tr[pl])
                    full = (N // m) * m
                    v = np.where(i < np.uint64(full), (np.uint64(base) + np.uint64(37) * (i % np.uint64(m))) % np.uint64(1000),
                                 np.uint64((base + 999) % 1000))
                elif dist == 4:
                    hot = (int(tr[pl]) + 500) % 1000 if int(d[pl][0]) == int(tr[pl]) else int(d[pl][0])
                    t0 = np.uint64((int(q_num[pl]) * 429496729) & 0xFFFFFFFF)
                    x = (hi - t0) & _MASK32
                    j = np.minimum(x // T5, np.uint64(3)).astype(np.int64)
                    dj = d[pl].astype(np.uint64).copy()
                    dj[0] = np.uint64(tr[pl])
                    v = np.where(hi < t0, np.uint64(hot), np.where(x < np.uint64(4) * T5, dj[j], uv))
                elif dist == 5:
                    v = np.full(N, (int(tr[pl]) + 500) % 1000, dtype=np.uint64)
                else:
                    raise ValueError(f"unknown dist {dist}")
                answers[pl, b] = v.astype(np.int32)
                if want_tokens:
                    tokens[pl, b] = (np.uint64(100) + _mulhi32(mix64(u ^ G) >> np.uint64(32), 11901)).astype(np.int32)
    return answers, tokens, tr.astype(np.int32)
