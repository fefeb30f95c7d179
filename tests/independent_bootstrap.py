"""A SECOND implementation of the problem-level bootstrap and of pass@k's integer input (a8), written from
the TEXT of include/scvote.h alone.

Why it exists (VERDICT r5 "What's missing" #4): a8 / a9 are new semantics -- the reference only has the axis label
`plot_helpers.py:21` -- so nothing in /root/reference can pin them ("parity unpinned", permanently).  Until round 6 the HIP
kernels were checked against oracle/scv_oracle.c only: one author, one reading of the specification.  This module is the other
reading: vectorised numpy, no loop over draws, written against

    include/scvote.h:187-196   "For resample r in [r_begin, r_end): draw P problem indices
                                idx_j = mulhi32(hi32(mix64(seed + G*(r*P+j+1))), P) and count, per budget, hits by tie class.
                                counts_out int64 [r_end-r_begin, B, M]; a drawn hit with n_modes >= M returns SCV_ERR_ARG"
    include/scvote.h:227       "mix64 = splitmix64 finaliser, G = 0x9E3779B97F4A7C15, mulhi32(a, n) = (a * n) >> 32"
    include/scvote.h:66-79     scv_cell: score = hit ? 1 / n_modes : 0; truth_count = histogram[truth]

and NOT against oracle/scv_oracle.c (it was not opened while this was written; the splitmix64 finaliser is Steele, Lea &
Flood 2014 / Vigna's public-domain splitmix64.c: z ^= z >> 30; z *= 0xBF58476D1CE4E5B9; z ^= z >> 27; z *= 0x94D049BB133111EB;
z ^= z >> 31).  It is test infrastructure: only tests/ import it.
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
ERR_ARG = -2001


def mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64's output function on a uint64 array (wrapping arithmetic)."""
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def draw_indices(P: int, r_begin: int, r_end: int, seed: int) -> np.ndarray:
    """idx[r - r_begin, j] for j in 0..P-1: the P problems resample r draws (with replacement)."""
    r = np.arange(r_begin, r_end, dtype=np.uint64)[:, None]
    j = np.arange(P, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        counter = r * np.uint64(P) + j + np.uint64(1)                   # r*P + j + 1, mod 2^64
        u = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + GOLDEN * counter)
    hi32 = u >> np.uint64(32)
    return ((hi32 * np.uint64(P)) >> np.uint64(32)).astype(np.int64)   # mulhi32(hi32, P): hi32 < 2^32 and P < 2^32, no overflow


def bootstrap(hit: np.ndarray, n_modes: np.ndarray, r_begin: int, r_end: int, seed: int, M: int):
    """hit, n_modes: [P, B] integer arrays (the two fields of scv_cell the bootstrap reads).  Returns (rc, counts int64 [R, B, M]):
    counts[r, b, m] = #{j : hit[idx_rj, b] and n_modes[idx_rj, b] == m}; rc = SCV_ERR_ARG when a DRAWN hit has n_modes >= M."""
    hit = np.asarray(hit).astype(bool)
    n_modes = np.asarray(n_modes).astype(np.int64)
    P, B = hit.shape
    R = r_end - r_begin
    out = np.zeros((R, B, M), dtype=np.int64)
    if P == 0 or R <= 0:
        return 0, out
    rc = 0
    # one class code per cell: the tie class of a hit, M (an overflow bucket) for a hit that does not fit, M + 1 for "no hit"
    code = np.where(hit, np.where(n_modes < M, n_modes, M), M + 1)
    step = max(1, (1 << 22) // max(P, 1))                               # resamples per block (bounds the [rows, P] temporaries)
    for r0 in range(r_begin, r_end, step):
        r1 = min(r_end, r0 + step)
        idx = draw_indices(P, r0, r1, seed)                             # [rows, P]
        rows = np.arange(r1 - r0)[:, None]
        for b in range(B):
            drawn = code[:, b][idx]                                     # [rows, P]
            flat = np.bincount((rows * (M + 2) + drawn).ravel(), minlength=(r1 - r0) * (M + 2)).reshape(r1 - r0, M + 2)
            out[r0 - r_begin:r1 - r_begin, b, :] = flat[:, :M]
            if flat[:, M].any():
                rc = ERR_ARG
    return rc, out


def truth_count(answers: np.ndarray, truth: np.ndarray, n_valid=None) -> np.ndarray:
    """c of the pass@k estimator (include/scvote.h:70): votes of cell (p, b) equal to truth[p], over its valid prefix."""
    answers = np.asarray(answers)
    P, B, N = answers.shape
    eq = answers == np.asarray(truth)[:, None, None]
    if n_valid is not None:
        nv = np.minimum(np.asarray(n_valid, dtype=np.int64), N)
        eq = eq & (np.arange(N)[None, None, :] < nv[None, :, None])
    return eq.sum(axis=2).astype(np.int64)


def accuracy_of_resamples(counts: np.ndarray, P: int) -> np.ndarray:
    """[R, B] float64: the a4 statistic of each resample, sum over tie classes m >= 1 of counts / m, over P (o1.py:210, 244)."""
    R, B, M = counts.shape
    acc = np.zeros((R, B), dtype=np.float64)
    for m in range(1, M):
        acc += counts[:, :, m] / m
    return acc / P
