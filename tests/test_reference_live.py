"""Integration against the LIVE, unmodified reference (build container only; skipped on the GPU
box where /root/reference does not exist).  Installs the drop-in into the imported o1 module (with
the oracle adapter as the engine -- CPU) and checks the reference's own drivers + plot/log writers
produce byte-identical results_log_*.json."""
import os
import random

import pytest

from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present")


PNGS = ("accuracy_vs_tokens_no_shade_regions.png", "token_limit_vs_actual.png", "just_ask_nicely_tokens.png")


def _image_hash(path):
    """sha256 over (shape, decoded RGBA pixels): independent of PNG metadata chunks."""
    import hashlib
    import matplotlib.image as mpimg
    import numpy as np
    img = np.ascontiguousarray(mpimg.imread(path))
    return hashlib.sha256(repr(img.shape).encode() + img.tobytes()).hexdigest()


def _pipeline_inputs(seed=99):
    rng = random.Random(seed)
    truths = [rng.randrange(1000) for _ in range(30)]
    samples = []
    for p in range(30):
        for T in [2 ** i for i in range(4, 11)]:
            samples.append((p, T, 0, truths[p] if rng.random() < 0.4 else rng.randrange(1000), rng.randrange(100, 3000)))
        pool = [truths[p] if rng.random() < 0.6 else rng.choice([1, 2]) for _ in range(8)]
        # keep tie sizes dyadic so the reference's accumulated float is order-independent
        for idx in range(8):
            samples.append((p, 2048, idx, pool[idx], rng.randrange(1500, 12000)))
    return truths, samples


@pytest.mark.parametrize("batched", [True, False])
def test_installed_dropin_reproduces_reference_logs(batched, oracle_engine):
    from o1_inference_scaling_laws_amd import o1_dropin
    consts = rh.reference_constants()
    truths, samples = _pipeline_inputs()
    ds = rh.make_dataset([str(t) for t in truths])
    cache = rh.build_cache(consts, ds, samples)
    with rh.imported_reference(ds, cache) as (o1, workdir):
        want = {n: open(os.path.join(workdir, "helpers", n)).read()
                for n in ("results_log_majority_vote.json", "results_log_just_ask_nicely.json")}
        for n in want:
            os.remove(os.path.join(workdir, "helpers", n))
        # the PNGs the reference's own plot_* functions rendered from ITS records
        # (plot_helpers.py:41-43, 56-57, 79-88; Agg backend, the reference's fixed figsize / dpi)
        want_png = {n: _image_hash(os.path.join(workdir, "graphs", n)) for n in PNGS}
        for n in PNGS:
            os.remove(os.path.join(workdir, "graphs", n))
        import contextlib, io
        o1_dropin.install(o1, engine=oracle_engine, batched=batched)
        with contextlib.redirect_stdout(io.StringIO()):
            o1.run_majority_vote_inference_experiments(ds, cache)
            o1.run_just_ask_nicely_experiments(ds, cache)
        for n, text in want.items():
            assert open(os.path.join(workdir, "helpers", n)).read() == text, n
        for n, h in want_png.items():
            assert _image_hash(os.path.join(workdir, "graphs", n)) == h, f"{n}: rendered pixels differ"


def test_shade_regions_and_full_range_records_match(oracle_engine):
    """The reference's non-default driver modes: shade_regions=True (o1.py:266-267: budgets up to 2^18,
    N up to 128 from the 2048 pool) and run_full_range=True (o1.py:298-299: 2^0..2^19, N = 1).  The
    records each driver hands to its plot function are captured for the unmodified module and for the
    installed drop-in and compared."""
    import contextlib
    import io
    from o1_inference_scaling_laws_amd import o1_dropin
    consts = rh.reference_constants()
    rng = random.Random(5)
    truths = [rng.randrange(1000) for _ in range(30)]
    samples = []
    for p in range(30):
        for T in [2 ** i for i in range(0, 11)] + [2 ** i for i in range(12, 20)]:
            samples.append((p, T, 0, truths[p] if rng.random() < 0.5 else rng.randrange(1000), rng.randrange(100, 3000)))
        q = rng.choice([0.2, 0.45, 0.7])
        for idx in range(128):
            r = rng.random()
            ans = truths[p] if r < q else (rng.choice([3, 4, 5]) if r < q + 0.4 else rng.randrange(1000))
            samples.append((p, 2048, idx, ans, rng.randrange(1500, 12000)))
    ds = rh.make_dataset([str(t) for t in truths])
    cache = rh.build_cache(consts, ds, samples)
    with rh.imported_reference(ds, cache) as (o1, workdir):
        captured = {}
        o1.plot_majority_vote_graph = lambda results, shade: captured.setdefault(("maj", shade), list(results))
        o1.plot_just_ask_nicely_graph = lambda results, full: captured.setdefault(("ask", full), list(results))
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            o1.run_majority_vote_inference_experiments(ds, cache, shade_regions=True)
            o1.run_just_ask_nicely_experiments(ds, cache, run_full_range=True)
        want = dict(captured)
        captured.clear()
        o1_dropin.install(o1, engine=oracle_engine, batched=True)
        o1.run_majority_vote_inference_experiments(ds, cache, shade_regions=True)
        o1.run_just_ask_nicely_experiments(ds, cache, run_full_range=True)
        got = dict(captured)
    assert set(got) == set(want) == {("maj", True), ("ask", True)}
    assert [r["token_limit"] for r in want[("maj", True)]] == [2 ** i for i in range(4, 19)]
    assert [r["token_limit"] for r in want[("ask", True)]] == [2 ** i for i in range(20)]
    for key in want:
        assert len(got[key]) == len(want[key])
        for g, w in zip(got[key], want[key]):
            assert list(g) == list(w) == ["token_limit", "accuracy", "avg_tokens_used"]
            assert g["token_limit"] == w["token_limit"]
            assert abs(g["accuracy"] - w["accuracy"]) < 1e-12          # o1.py:239 accumulates in completion order
            assert repr(float(g["avg_tokens_used"])) == repr(float(w["avg_tokens_used"]))
            assert type(g["avg_tokens_used"]) is type(w["avg_tokens_used"])
