"""Integration against the LIVE, unmodified reference (build container only; skipped on the GPU
box where /root/reference does not exist).  Installs the drop-in into the imported o1 module (with
the oracle adapter as the engine -- CPU) and checks the reference's own drivers + plot/log writers
produce byte-identical results_log_*.json."""
import os
import random

import pytest

from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="/root/reference not present")


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
        import contextlib, io
        o1_dropin.install(o1, engine=oracle_engine, batched=batched)
        with contextlib.redirect_stdout(io.StringIO()):
            o1.run_majority_vote_inference_experiments(ds, cache)
            o1.run_just_ask_nicely_experiments(ds, cache)
        for n, text in want.items():
            assert open(os.path.join(workdir, "helpers", n)).read() == text, n
        assert os.path.exists(os.path.join(workdir, "graphs", "accuracy_vs_tokens_no_shade_regions.png"))
