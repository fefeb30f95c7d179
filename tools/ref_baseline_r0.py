#!/usr/bin/env python3
"""R0 baseline: the UNMODIFIED reference's run_experiments (o1.py:216-247) on a warm
in-memory synthetic cache, save_cache no-op'd so only the vote loop is timed.  Runs only where
/root/reference exists (the build container); the number is quoted in DESIGN.md as container-measured."""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh

consts = rh.reference_constants()
rng = random.Random(1)
truths = [rng.randrange(1000) for _ in range(30)]
ds = rh.make_dataset([str(t) for t in truths])
boot = [(p, T, 0, truths[p], 100) for p in range(30) for T in [2 ** i for i in range(4, 11)]] + \
       [(p, 2048, i, truths[p], 100) for p in range(30) for i in range(8)]
out = {"cpu_count": os.cpu_count(), "results": []}
with rh.imported_reference(ds, rh.build_cache(consts, ds, boot)) as (o1, workdir):
    o1.save_cache = lambda cache, filename: None
    for N in (256, 2048):
        samples = [(p, 2048, i, truths[p] if rng.random() < 0.5 else rng.randrange(1000), rng.randrange(100, 12000))
                   for p in range(30) for i in range(N)]
        cache = rh.build_cache(consts, ds, samples)
        t0 = time.perf_counter()
        acc, avg = o1.run_experiments(ds, cache, 2048, N)
        dt = time.perf_counter() - t0
        out["results"].append({"P": 30, "N": N, "seconds": dt, "votes_per_s": 30 * N / dt, "accuracy": acc})
        print(f"R0 unmodified o1.run_experiments P=30 N={N}: {dt:.2f} s = {30 * N / dt:.0f} votes/s", flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_ref_baseline_r0_container.json"), "w"), indent=1)
