"""a9 / a8 against a SECOND, independent implementation.

The reference has neither a bootstrap nor pass@k (only the axis label helpers/plot_helpers.py:21), so these rows stay "parity
unpinned" by the reference.  What can be done is remove the single-author risk: tests/independent_bootstrap.py restates the
bootstrap in vectorised numpy from the text of include/scvote.h:187-196 alone.  Here (CPU) it is compared, whole tables, with
oracle/scv_oracle.c's scvo_bootstrap -- two readings of one specification must agree -- and the resample statistics are checked
for the properties ANY correct problem-level bootstrap has (mean of the resample accuracies -> the accuracy; the percentile
interval brackets it; slices of the resample range concatenate; the draws are uniform over the problems).  The `-m gpu` half
compares the HIP kernels with the numpy table on C5-sized cell tables (tests below the marker)."""
import math

import numpy as np
import pytest

from oracle import coracle
from tests import independent_bootstrap as ib


def _cells(P, B, rng, p_hit=0.6, max_class=4):
    c = np.zeros((P, B), dtype=coracle.CELL_DTYPE)
    c["n_modes"] = rng.integers(1, max_class + 1, size=(P, B))
    c["hit"] = rng.random((P, B)) < p_hit
    c["max_count"] = 5
    c["truth_count"] = np.where(c["hit"] == 1, 5, rng.integers(0, 5, size=(P, B)))
    return c


def test_mix64_is_the_published_splitmix64():
    """Known answers of splitmix64 (Vigna's splitmix64.c, seed 0: the first outputs are mix64(G), mix64(2G), mix64(3G))."""
    g = 0x9E3779B97F4A7C15
    got = ib.mix64(np.array([g, (2 * g) & (2**64 - 1), (3 * g) & (2**64 - 1)], dtype=np.uint64))
    assert [int(x) for x in got] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


@pytest.mark.parametrize("P,B,R", [(1, 1, 9), (2, 3, 40), (30, 8, 100), (257, 4, 200), (10000, 1, 60), (4099, 2, 33)])
def test_numpy_table_equals_the_c_oracle_table(P, B, R):
    rng = np.random.default_rng(P * 131 + B)
    cells = _cells(P, B, rng)
    for (r0, seed, M) in ((0, 0xB007, 5), (7, 2**64 - 3, 9), (R // 2, 0, 1025)):
        rc_c, want = coracle.bootstrap(cells, r0, r0 + R, seed, M)
        rc_n, got = ib.bootstrap(cells["hit"], cells["n_modes"], r0, r0 + R, seed, M)
        assert rc_c == 0 and rc_n == 0
        assert got.shape == want.shape == (R, B, M) and np.array_equal(got, want)
        assert (got[:, :, 0] == 0).all() and (got.sum(axis=2) <= P).all()


def test_overflow_of_the_class_bound_is_an_error_in_both():
    rng = np.random.default_rng(5)
    cells = _cells(500, 2, rng, p_hit=0.9, max_class=4)
    rc_c, _ = coracle.bootstrap(cells, 0, 20, 1, 4)          # class 4 is present among the hits: M = 4 does not hold it
    rc_n, _ = ib.bootstrap(cells["hit"], cells["n_modes"], 0, 20, 1, 4)
    assert rc_c != 0 and rc_n == ib.ERR_ARG
    cells["hit"][cells["n_modes"] == 4] = 0                  # a class that is present but never a HIT does not overflow
    rc_c, want = coracle.bootstrap(cells, 0, 20, 1, 4)
    rc_n, got = ib.bootstrap(cells["hit"], cells["n_modes"], 0, 20, 1, 4)
    assert rc_c == 0 and rc_n == 0 and np.array_equal(got, want)


@pytest.mark.parametrize("G", [1, 2, 3, 5, 8])
def test_slices_of_the_resample_range_concatenate(G):
    """C5 on G GPUs gives rank g the resamples [g R / G, (g+1) R / G): the slices must be the one table, cut."""
    rng = np.random.default_rng(G)
    cells = _cells(1000, 2, rng)
    R = 203
    _, whole = ib.bootstrap(cells["hit"], cells["n_modes"], 0, R, 99, 6)
    parts = [ib.bootstrap(cells["hit"], cells["n_modes"], g * R // G, (g + 1) * R // G, 99, 6)[1] for g in range(G)]
    assert np.array_equal(np.concatenate(parts, axis=0), whole)
    parts_c = [coracle.bootstrap(cells, g * R // G, (g + 1) * R // G, 99, 6)[1] for g in range(G)]
    assert np.array_equal(np.concatenate(parts_c, axis=0), whole)


def test_the_draws_are_uniform_over_the_problems():
    """idx = mulhi32(hi32(mix64(.)), P): every problem is drawn ~ R times in R resamples of P draws (chi-square, 5 sigma)."""
    P, R = 1000, 400
    idx = ib.draw_indices(P, 0, R, 12345)
    assert idx.shape == (R, P) and idx.min() >= 0 and idx.max() < P
    counts = np.bincount(idx.ravel(), minlength=P)
    chi2 = ((counts - R) ** 2 / R).sum()
    assert abs(chi2 - (P - 1)) < 5 * math.sqrt(2 * (P - 1))
    assert len({tuple(row[:16]) for row in idx}) == R         # no two resamples repeat their first draws


def test_statistics_of_the_resample_accuracies():
    """What any correct problem-level bootstrap satisfies: the mean of the R resample accuracies is the accuracy within
    3 sigma / sqrt(R) (sigma: the standard error of the accuracy = std of the per-problem scores / sqrt(P)), their spread is
    that standard error, and the 2.5 / 97.5 percentile interval brackets the accuracy."""
    from o1_inference_scaling_laws_amd import scoring
    rng = np.random.default_rng(77)
    P, B, R, M = 10000, 3, 1000, 5
    cells = _cells(P, B, rng, p_hit=0.55)
    rc, counts = ib.bootstrap(cells["hit"], cells["n_modes"], 0, R, 0xB007, M)
    assert rc == 0
    score = np.where(cells["hit"] == 1, 1.0 / cells["n_modes"], 0.0)          # o1.py:206-210
    acc = score.mean(axis=0)
    se = score.std(axis=0) / math.sqrt(P)
    boot_acc = ib.accuracy_of_resamples(counts, P)
    assert (np.abs(boot_acc.mean(axis=0) - acc) < 3 * se / math.sqrt(R) + 1e-12).all()
    assert (np.abs(boot_acc.std(axis=0) / se - 1) < 0.1).all()
    lo, hi = np.percentile(boot_acc, [2.5, 97.5], axis=0)
    assert (lo < acc).all() and (acc < hi).all() and ((hi - lo) / (2 * 1.96 * se) > 0.85).all() and ((hi - lo) / (2 * 1.96 * se) < 1.15).all()
    # the product's host function over the same counters gives the same numbers
    acc_p, lo_p, hi_p = scoring.bootstrap_percentiles(counts, P)
    assert np.allclose(acc_p, boot_acc, rtol=0, atol=1e-15) and np.allclose(lo_p, lo) and np.allclose(hi_p, hi)


def test_truth_count_restatement_on_the_oracle():
    """a8's integer input: c = votes equal to the truth over the valid prefix, restated in three numpy lines, against the oracle's."""
    a, _, tr = coracle.synth_fill(40, 5, 300, 3, 1)
    nv = np.array([300, 1, 0, 77, 1000], dtype=np.int32)
    for n_valid in (None, nv):
        want = coracle.aggregate(a, tr, n_valid=n_valid)["cells"]["truth_count"]
        assert np.array_equal(ib.truth_count(a, tr, n_valid), want)


# ---- the HIP kernels against the independent table ---------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("boot_path", [0, 3])
@pytest.mark.parametrize("P,B,M", [(10000, 1, 4), (10000, 8, 6), (1250, 1, 1025), (70000, 2, 3)])
def test_kernel_table_equals_the_independent_numpy_table(hip_engine, boot_path, P, B, M):
    """C5-sized cell tables (P = 10^4, B = 1; and B = 8, a table beyond the LDS form, every class bound): the WHOLE
    [R, B, M] table of scv_bootstrap == the numpy restatement == scvo_bootstrap, for both bootstrap kernels."""
    rng = np.random.default_rng(P + B)
    cells = _cells(P, B, rng, max_class=min(M - 1, 5))
    hip_engine.set_option("boot_path", boot_path)
    try:
        for (r0, r1, seed) in ((0, 1000 if P * B <= 20000 else 120, 0xB007 ^ 20240914), (3, 64, 2**64 - 1)):
            got = hip_engine.bootstrap(cells, r0, r1, seed, M)
            rc, want = ib.bootstrap(cells["hit"], cells["n_modes"], r0, r1, seed, M)
            assert rc == 0 and got.shape == want.shape and np.array_equal(got, want)
            rc, third = coracle.bootstrap(cells, r0, r1, seed, M)
            assert rc == 0 and np.array_equal(third, want)
    finally:
        hip_engine.set_option("boot_path", 0)


@pytest.mark.gpu
def test_vote_plus_bootstrap_in_one_launch_equals_the_independent_table(hip_engine):
    """The C5 form (vote + bootstrap in ONE launch): cells from the kernel, resample table against numpy on those cells, the
    truth counts against the three-line restatement, the slices R/G of G = 1 .. 8 ranks against the whole."""
    import torch
    from o1_inference_scaling_laws_amd.engine import cells_from_torch
    P, B, N, R = 2000, 1, 1 << 14, 500
    dev = torch.device("cuda:0")
    ans = torch.empty((P, B, N), dtype=torch.int32, device=dev)
    tr = torch.empty((P,), dtype=torch.int32, device=dev)
    hip_engine.synth_fill_device(ans, None, tr, P=P, B=B, N=N, seed=61, dist=3)
    _, cells, _, boot = hip_engine.aggregate_bootstrap_device(ans, tr, 0, R, 4242, 4)
    hip_engine.sync()
    c = cells_from_torch(cells)
    rc, want = ib.bootstrap(c["hit"], c["n_modes"], 0, R, 4242, 4)
    assert rc == 0 and np.array_equal(boot.cpu().numpy(), want)
    assert np.array_equal(c["truth_count"], ib.truth_count(ans.cpu().numpy(), tr.cpu().numpy()))
    for G in (2, 3, 8):
        parts = [hip_engine.bootstrap(c, g * R // G, (g + 1) * R // G, 4242, 4) for g in range(G)]
        assert np.array_equal(np.concatenate(parts, axis=0), want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3000, 3, 1), (2000, 2, 4), (1500, 4, 8), (800, 3, 48), (500, 2, 96), (400, 3, 128), (300, 2, 1000), (64, 2, 5000), (12, 2, 70000)])
def test_truth_count_of_every_kernel_family_equals_the_restatement(hip_engine, shape):
    """a8's integer input from every regime of the dispatch (few votes / sorted / register-resident / dense / streaming), ragged budgets."""
    P, B, N = shape
    a, _, tr = coracle.synth_fill(P, B, N, 17, 4)
    nv = np.array([max(1, N >> (B - 1 - b)) for b in range(B)], dtype=np.int32)
    for n_valid in (None, nv):
        got = hip_engine.aggregate(a, tr, n_valid=n_valid)
        assert np.array_equal(got.cells["truth_count"], ib.truth_count(a, tr, n_valid))
