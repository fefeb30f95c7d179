"""Property tests (hypothesis) of the oracle and the host logic against statistics.multimode -- the
stdlib function the reference calls at o1.py:202.  CPU only."""
import statistics
from fractions import Fraction

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from o1_inference_scaling_laws_amd import extract, scoring
from oracle import coracle, pyoracle

bins = st.integers(min_value=0, max_value=1023)
anyint = st.one_of(st.integers(min_value=-5, max_value=1005), st.sampled_from([-10 ** 12, 10 ** 9, 2 ** 40, 1000, 1023, 1024]))


@settings(max_examples=200, deadline=None)
@given(votes=st.lists(bins, min_size=0, max_size=60), truth=st.integers(min_value=-2, max_value=1026))
def test_c_oracle_cell_equals_multimode(votes, truth):
    a = np.array(votes, dtype=np.int32).reshape(1, 1, len(votes))
    out = coracle.aggregate(a, np.array([truth], dtype=np.int32))
    c = out["cells"][0, 0]
    modes = statistics.multimode(votes)
    assert c["n_modes"] == len(modes)
    assert c["max_count"] == (votes.count(modes[0]) if modes else 0)
    assert c["min_mode"] == (min(modes) if modes else -1)
    assert c["truth_count"] == votes.count(truth)
    assert bool(c["hit"]) == (truth in modes)
    score = (1 / len(modes)) if truth in modes else 0
    assert pyoracle.process_votes(votes, [0] * len(votes), truth)[0] == score
    tie = out["tie_class_hits"][0]
    assert tie.sum() == int(c["hit"]) and (not c["hit"] or tie[len(modes)] == 1)


@settings(max_examples=150, deadline=None)
@given(votes=st.lists(anyint, min_size=1, max_size=30), truth=anyint)
def test_domain_encoding_preserves_multimode_semantics(votes, truth):
    """Arbitrary Python ints -> spare-bin codes: mode multiplicities and truth membership survive."""
    enc = extract.ProblemEncoder()
    try:
        t = enc.encode(truth)
        coded = [enc.encode(v) for v in votes]
    except extract.DomainOverflow:
        return
    want = pyoracle.cell_integers(votes, truth)
    got = coracle.aggregate(np.array(coded, dtype=np.int32).reshape(1, 1, -1), np.array([t], dtype=np.int32))["cells"][0, 0]
    assert (got["max_count"], got["truth_count"], got["n_modes"], got["hit"]) == (
        want["max_count"], want["truth_count"], want["n_modes"], want["hit"])


@settings(max_examples=100, deadline=None)
@given(data=st.data())
def test_prefix_budgets_equal_recount_from_scratch(data):
    """The running-histogram view of prefix budgets == recounting every prefix (what the dense path does)."""
    pool = data.draw(st.lists(st.integers(min_value=0, max_value=6), min_size=1, max_size=40))
    truth = data.draw(st.integers(min_value=0, max_value=6))
    cuts = data.draw(st.lists(st.integers(min_value=0, max_value=len(pool)), min_size=1, max_size=6))
    a = np.array(pool, dtype=np.int32)
    dense = np.ascontiguousarray(np.broadcast_to(a, (1, len(cuts), len(pool))))
    out = coracle.aggregate(dense, np.array([truth], dtype=np.int32), n_valid=np.array(cuts, dtype=np.int32))
    for b, n in enumerate(cuts):
        modes = statistics.multimode(pool[:n])
        assert out["cells"][0, b]["n_modes"] == len(modes) and bool(out["cells"][0, b]["hit"]) == (truth in modes)


@settings(max_examples=100, deadline=None)
@given(hits=st.lists(st.tuples(st.integers(min_value=1, max_value=12), st.integers(min_value=0, max_value=40)), max_size=6),
       P=st.integers(min_value=1, max_value=500))
def test_accuracy_float_is_the_rational_rounded(hits, P):
    tie = np.zeros(1025, dtype=np.int64)
    for m, c in hits:
        tie[m] += c
    exact = scoring.exact_accuracy_from_tie_classes(tie, P)
    assert abs(Fraction(scoring.accuracy_from_tie_classes(tie, P)) - exact) <= Fraction(1, 10 ** 13) * max(1, exact)
    if all(m & (m - 1) == 0 for m, _ in hits):                 # dyadic tie sizes: exactly representable sums
        assert scoring.accuracy_from_tie_classes(tie, P) == float(sum(Fraction(int(tie[m]), m) for m in range(1, 1025))) / P


def _online_prefix_modes(votes, truth):
    """Python model of scv_lane_prefix (csrc/scvote_kernels.hip.h): votes enter in order; vote i of value x has count
    c = #{ j <= i : x_j == x }, and adding it changes the mode statistics exactly one way.  Yields the state after
    every vote: (max_count, n_modes, min_mode, truth_count)."""
    maxc = n_modes = tc = 0
    min_mode = None
    for i, x in enumerate(votes):
        c = sum(1 for j in range(i + 1) if votes[j] == x)           # the kernel: packed compares with the earlier votes
        if c > maxc:
            maxc, n_modes, min_mode = c, 1, x
        elif c == maxc:
            n_modes, min_mode = n_modes + 1, min(min_mode, x)
        tc += x == truth
        yield maxc, n_modes, min_mode, tc


@settings(max_examples=300, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=12), min_size=1, max_size=64), st.integers(min_value=0, max_value=12))
def test_online_mode_tracking_equals_multimode_on_every_prefix(votes, truth):
    """The one-pass prefix kernel's invariant: the running state after vote i IS statistics.multimode's answer for the
    prefix 0..i (o1.py:202-213 applied to samples[:i+1]) -- so a budget is a snapshot, whatever its boundary."""
    for i, (maxc, n_modes, min_mode, tc) in enumerate(_online_prefix_modes(votes, truth)):
        modes = statistics.multimode(votes[: i + 1])
        assert n_modes == len(modes) and min_mode == min(modes) and maxc == votes[: i + 1].count(modes[0])
        assert tc == votes[: i + 1].count(truth)
        assert (tc == maxc and maxc > 0) == (truth in modes)                     # the kernel's `hit`


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=1023), min_size=1, max_size=200))
def test_sixteen_bit_bin_packing_of_the_register_kernels(votes):
    """Python model of the 16-bit histogram of scv_reg_cells (G = 16): bin b lives at byte address A = KB - 2 b; a vote adds
    `alignbyte(1, 1, A)` = 1 << 16 * (bit 1 of A) to the 32-bit WORD at A & ~3, a count is read back as the 16-bit value
    at A.  Counts stay below 2^16, so the two bins of a word never disturb each other."""
    KB = 2 * 1023 + 4096                                     # any 4-byte aligned base works
    words = {}
    for v in votes:
        A = KB - 2 * v
        inc = ((1 << 32 | 1) >> (8 * (A & 3))) & 0xFFFFFFFF  # v_alignbyte_b32 D = ({hi = 1, lo = 1} >> 8 * S2[1:0])
        assert inc == (1 << (16 * ((A >> 1) & 1)))
        words[A & ~3] = (words.get(A & ~3, 0) + inc) & 0xFFFFFFFF
    for b in set(votes):
        A = KB - 2 * b
        w = words[A & ~3]
        assert (w >> (8 * (A & 3))) & 0xFFFF == votes.count(b)                   # ds_read_u16 at A (little endian)
