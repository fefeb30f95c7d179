"""Integration against the LIVE, unmodified reference.  Installs the drop-in into the imported o1 module
and checks that the reference's own drivers + plot/log writers produce byte-identical results_log_*.json
and pixel-identical PNGs.

* CPU tests (build container): the oracle adapter is the engine; the reference is imported from
  /root/reference, and once more from the compiled copy in oracle/_ref (oracle/make_ref.py) to show that
  both are the same program.
* ``-m gpu`` tests (GPU box, where /root/reference does not exist): the HIP ``Engine`` is the engine and the
  reference is the bytecode in oracle/_ref that ``__graft_entry__.build()`` compiled from the unmodified
  sources -- the unmodified reference and the HIP engine in ONE process.  Missing bytecode FAILS there."""
import os
import random

import pytest

from oracle import ref_harness as rh

needs_reference = pytest.mark.skipif(not rh.reference_available(), reason="neither /root/reference nor oracle/_ref present")


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


def _check_installed_dropin_reproduces_reference_logs(batched, engine):
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
        o1_dropin.install(o1, engine=engine, batched=batched)
        with contextlib.redirect_stdout(io.StringIO()):
            o1.run_majority_vote_inference_experiments(ds, cache)
            o1.run_just_ask_nicely_experiments(ds, cache)
        for n, text in want.items():
            assert open(os.path.join(workdir, "helpers", n)).read() == text, n
        for n, h in want_png.items():
            assert _image_hash(os.path.join(workdir, "graphs", n)) == h, f"{n}: rendered pixels differ"


def _check_shade_regions_and_full_range_records_match(engine):
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
        o1_dropin.install(o1, engine=engine, batched=True)
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


# ---- CPU: oracle adapter as the engine ------------------------------------------------------------------------

@needs_reference
@pytest.mark.parametrize("batched", [True, False])
def test_installed_dropin_reproduces_reference_logs(batched, oracle_engine):
    _check_installed_dropin_reproduces_reference_logs(batched, oracle_engine)


@needs_reference
def test_shade_regions_and_full_range_records_match(oracle_engine):
    _check_shade_regions_and_full_range_records_match(oracle_engine)


@pytest.mark.skipif(not os.path.isfile("/root/reference/o1.py"), reason="/root/reference not present")
def test_compiled_reference_in_oracle_ref_is_the_same_program(oracle_engine, monkeypatch):
    """oracle/make_ref.py: the bytecode under oracle/_ref is built from the sources whose hashes its manifest
    records, carries the same constants, and -- imported sourceless, as on the GPU box -- reproduces the logs and
    PNGs the drop-in is compared with."""
    import hashlib
    from oracle import make_ref
    assert make_ref.build() == make_ref.REF_OUT and make_ref.available()
    m = make_ref.manifest()
    for f, h in m["sha256"].items():
        assert hashlib.sha256(open(os.path.join("/root/reference", f), "rb").read()).hexdigest() == h
    assert m["constants"] == rh.reference_constants()
    assert not any(n.endswith(".py") for _, _, names in os.walk(make_ref.REF_OUT) for n in names), "sources must not be copied"
    monkeypatch.setattr(rh, "REFERENCE_SOURCES", "/nonexistent")
    assert rh.reference_kind() == "bytecode" and rh.reference_constants() == m["constants"]
    _check_installed_dropin_reproduces_reference_logs(True, oracle_engine)


# ---- GPU: the unmodified reference and the HIP engine in one process -------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("batched", [True, False])
def test_gpu_installed_dropin_reproduces_reference_logs(batched, hip_engine):
    assert rh.reference_available(), "oracle/_ref is missing on the GPU box: __graft_entry__.build() must run where /root/reference exists"
    _check_installed_dropin_reproduces_reference_logs(batched, hip_engine)


@pytest.mark.gpu
def test_gpu_shade_regions_and_full_range_records_match(hip_engine):
    assert rh.reference_available(), "oracle/_ref is missing on the GPU box"
    _check_shade_regions_and_full_range_records_match(hip_engine)


@pytest.mark.gpu
def test_gpu_default_engine_installed_without_arguments(hip_engine):
    """INTEGRATION.md section 1 verbatim: ``install(o1)`` with no engine argument creates the process-wide HIP engine;
    the reference's own ``run_experiments`` call sites (o1.py:277, :302) then reach the GPU."""
    from o1_inference_scaling_laws_amd import o1_dropin
    assert rh.reference_available()
    consts = rh.reference_constants()
    truths, samples = _pipeline_inputs(seed=7)
    ds = rh.make_dataset([str(t) for t in truths])
    cache = rh.build_cache(consts, ds, samples)
    with rh.imported_reference(ds, cache) as (o1, _workdir):
        want = [o1.run_experiments(ds, cache, 2048, n) for n in (1, 2, 4, 8)]
        want_one = o1.process_single_example(ds[3], 2048, cache, 8)
        cfg = o1_dropin.install(o1)
        assert type(cfg.get_engine()).__name__ == "Engine"
        got = [o1.run_experiments(ds, cache, 2048, n) for n in (1, 2, 4, 8)]
        got_one = o1.process_single_example(ds[3], 2048, cache, 8)
    for (ga, gt), (wa, wt) in zip(got, want):
        assert ga == wa and repr(gt) == repr(wt) and type(gt) is type(wt)     # dyadic tie sizes: accuracy is exact
    assert got_one == want_one and type(got_one[0]) is type(want_one[0])
