"""Drop-in replacements for the aggregation entry points of the reference, same names, same
argument meaning, same return types -- backed by the HIP engine.

    reference (/root/reference/o1.py)                      here
    :167 process_single_example(example, token_limit, cache, N) -> (score, total_tokens)
    :216 run_experiments(dataset, cache, token_limit, N)        -> (accuracy, avg_tokens_used)
    :250 run_majority_vote_inference_experiments(dataset, cache, shade_regions=False)
    :288 run_just_ask_nicely_experiments(dataset, cache, run_full_range=False)

``install(o1_module)`` rebinds the first two (and, batched, the last two) on the reference module:
its drivers look ``run_experiments`` up as a module global at call time (o1.py:277, :302), so the
``run_*`` / ``plot_*`` entry points and the ``results_log_*.json`` schema stay untouched.

The engine never talks to the network: a sample that is not in the cache is a failed sample and
votes (0, 0) exactly like the reference's own failure path (o1.py:190-192; see extract.py).
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass, field
from typing import Any, Callable

import numpy as np

from .extract import build_vote_tensors

_default_engine = None


def default_engine():
    """The process-wide HIP engine (created on first use; raises without the library / a GPU)."""
    global _default_engine
    if _default_engine is None:
        from .engine import Engine
        _default_engine = Engine()
    return _default_engine


@dataclass
class DropInConfig:
    """What the reference module contributes: key constants and its own side-effect functions."""
    model: str                                    # o1.py:17 O1_MODEL
    prompt: str                                   # o1.py:21-30 PROMPT (unformatted template)
    save_cache: Callable[[dict, str], None] | None = None          # o1.py:66
    cache_filename: str = "helpers/response_cache.json"            # o1.py:20
    plot_majority_vote_graph: Callable | None = None               # helpers/plot_helpers.py:9
    plot_just_ask_nicely_graph: Callable | None = None             # helpers/plot_helpers.py:64
    engine: Any = None                            # anything with .aggregate(answers, truth, tokens, n_valid)
    helper_folder: str = "helpers"                # plot_helpers.py:7
    extra: dict = field(default_factory=dict)
    # wall-clock split of the calls made through this config, in seconds (bench.py's cpu_baseline.dropin_loop reads it):
    #   extract  cache -> vote tensors (extract.build_vote_tensors: the reference's key scheme, o1.py:85-88,119)
    #   engine   the blocking HOST-mode engine call (staging H2D + kernel + results back)
    #   kernel   of that, the hot-path kernel itself (hipEvents; only when the engine was created with timing=True)
    #   floats   integer counters -> the reference's floats (scoring.py; o1.py:244-245)
    timings: dict = field(default_factory=lambda: {"extract": 0.0, "engine": 0.0, "kernel": 0.0, "floats": 0.0, "calls": 0, "votes": 0})

    def get_engine(self):
        return self.engine if self.engine is not None else default_engine()


def majority_vote_budgets(shade_regions: bool = False):
    """o1.py:266-276 -> [(token_limit, actual_token_limit, N)]."""
    token_limits = [2 ** i for i in range(4, 19)] if shade_regions else [2 ** i for i in range(4, 15)]
    out = []
    for token_limit in token_limits:
        actual_token_limit = min(2 ** 11, token_limit)
        out.append((token_limit, actual_token_limit, token_limit // actual_token_limit))
    return out


def just_ask_nicely_budgets(run_full_range: bool = False):
    """o1.py:297-302 -> [(token_limit, token_limit, 1)]."""
    token_limits = [2 ** i for i in range(20)] if run_full_range else [2 ** i for i in range(4, 12)]
    return [(t, t, 1) for t in token_limits]


def _extract(cfg: DropInConfig, eng, dataset, cache, budgets):
    """cache -> vote tensors, written straight into page-locked memory when the engine hands it out (Engine.pinned_empty:
    the HOST-mode call then DMAs the tensors in place instead of copying them through its bounce slots)."""
    t0 = time.perf_counter()
    vt = build_vote_tensors(dataset, cache, budgets, cfg.model, cfg.prompt, alloc=getattr(eng, "pinned_empty", None))
    cfg.timings["extract"] += time.perf_counter() - t0
    cfg.timings["votes"] += int(len(dataset)) * int(sum(n for _, n in budgets))
    return vt


def _timed_engine_call(cfg: DropInConfig, eng, fn, *args, **kwargs):
    drain = getattr(eng, "drain_kernel_ns", None) if getattr(eng, "timing", False) else None
    if drain is not None:
        drain()
    t0 = time.perf_counter()
    res = fn(*args, **kwargs)
    cfg.timings["engine"] += time.perf_counter() - t0
    cfg.timings["calls"] += 1
    if drain is not None:
        cfg.timings["kernel"] += drain()[0] * 1e-9
    return res


def _aggregate(cfg: DropInConfig, dataset, cache, budgets):
    eng = cfg.get_engine()
    vt = _extract(cfg, eng, dataset, cache, budgets)
    return _timed_engine_call(cfg, eng, eng.aggregate, vt.answers, vt.truth, tokens=vt.tokens, n_valid=vt.n_valid)


def _floats(cfg: DropInConfig, res, b: int):
    t0 = time.perf_counter()
    out = (res.accuracy(b), res.avg_tokens_used(b))
    cfg.timings["floats"] += time.perf_counter() - t0
    return out


def process_single_example(cfg: DropInConfig, example: dict, token_limit: int, cache: dict, N: int):
    """o1.py:167-213 -> (score, total_tokens): score is int 0 or float 1/len(modes)."""
    res = _aggregate(cfg, [example], cache, [(token_limit, N)])
    cell = res.cells[0, 0]
    score = 1 / int(cell["n_modes"]) if cell["hit"] else 0      # o1.py:204-210
    return score, int(res.cell_tokens[0, 0])


def run_experiments(cfg: DropInConfig, dataset, cache: dict, token_limit: int, N: int):
    """o1.py:216-247 -> (accuracy: float, avg_tokens_used: np.float64), incl. the save_cache side effect."""
    res = _aggregate(cfg, dataset, cache, [(token_limit, N)])
    if cfg.save_cache is not None:
        cfg.save_cache(cache, cfg.cache_filename)                # o1.py:242
    return _floats(cfg, res, 0)


def _run_family(cfg: DropInConfig, dataset, cache, budgets):
    """All budgets of a family in at most two engine calls.

    Budgets that share a key token limit vote over PREFIXES of one sample pool (o1.py:274-277 with the
    idx-keyed cache of o1.py:85-88: T = 2^11, 2^12, 2^13, ... are samples 0..N-1 of the 2048-token
    pool): they go through the engine's prefix mode as ONE pool [P, Nmax] + n_valid -- the pool is
    streamed once instead of being expanded B times.  The remaining budgets (one
    sample each, distinct key limits) go as one dense [P, B', 1] call.  An engine without
    ``aggregate_prefix`` gets everything as one dense [P, B, Nmax] call."""
    eng = cfg.get_engine()
    by_key = {}
    for i, (_token_limit, key_limit, _n) in enumerate(budgets):
        by_key.setdefault(key_limit, []).append(i)
    pooled = {k: idx for k, idx in by_key.items() if len(idx) > 1 and hasattr(eng, "aggregate_prefix")}
    dense = [i for k, idx in by_key.items() if k not in pooled for i in idx]
    floats = [None] * len(budgets)
    if dense:
        res = _aggregate(cfg, dataset, cache, [(budgets[i][1], budgets[i][2]) for i in dense])
        for j, i in enumerate(dense):
            floats[i] = _floats(cfg, res, j)
    for key_limit, idx in pooled.items():
        ns = [budgets[i][2] for i in idx]
        vt = _extract(cfg, eng, dataset, cache, [(key_limit, max(ns))])
        res = _timed_engine_call(cfg, eng, eng.aggregate_prefix, vt.answers[:, 0, :], vt.truth, np.asarray(ns, dtype=np.int32),
                                 tokens=vt.tokens[:, 0, :])
        for j, i in enumerate(idx):
            floats[i] = _floats(cfg, res, j)
    if cfg.save_cache is not None:
        cfg.save_cache(cache, cfg.cache_filename)
    results = []
    for i, (token_limit, _key_limit, _n) in enumerate(budgets):
        results.append({                                          # o1.py:278-283 / :303-307
            "token_limit": token_limit,
            "accuracy": floats[i][0],
            "avg_tokens_used": floats[i][1],
        })
    return results


def run_majority_vote_inference_experiments(cfg: DropInConfig, dataset, cache, shade_regions: bool = False):
    """o1.py:250-285, batched.  Hands the records to the reference's own plot function when the
    config has it; otherwise writes only the results log with the reference's schema."""
    results = _run_family(cfg, dataset, cache, majority_vote_budgets(shade_regions))
    if cfg.plot_majority_vote_graph is not None:
        cfg.plot_majority_vote_graph(results, shade_regions)      # o1.py:285
    elif not shade_regions:
        _write_results_log(cfg, "results_log_majority_vote.json", results)   # plot_helpers.py:59-60
    return results


def run_just_ask_nicely_experiments(cfg: DropInConfig, dataset, cache, run_full_range: bool = False):
    """o1.py:288-309, batched."""
    results = _run_family(cfg, dataset, cache, just_ask_nicely_budgets(run_full_range))
    if cfg.plot_just_ask_nicely_graph is not None:
        cfg.plot_just_ask_nicely_graph(results, run_full_range)   # o1.py:309
    elif not run_full_range:
        _write_results_log(cfg, "results_log_just_ask_nicely.json", results)  # plot_helpers.py:85-86
    return results


def _write_results_log(cfg: DropInConfig, name: str, results):
    os.makedirs(cfg.helper_folder, exist_ok=True)
    with open(os.path.join(cfg.helper_folder, name), "w") as f:
        json.dump(results, f, indent=2)


def config_from_module(o1_module, engine=None) -> DropInConfig:
    """Read the key constants and side-effect functions off the (imported) reference module."""
    return DropInConfig(
        model=o1_module.O1_MODEL,
        prompt=o1_module.PROMPT,
        save_cache=getattr(o1_module, "save_cache", None),
        cache_filename=getattr(o1_module, "RESPONSE_CACHE_FILENAME", "helpers/response_cache.json"),
        plot_majority_vote_graph=getattr(o1_module, "plot_majority_vote_graph", None),
        plot_just_ask_nicely_graph=getattr(o1_module, "plot_just_ask_nicely_graph", None),
        engine=engine,
    )


def install(o1_module, engine=None, batched: bool = True) -> DropInConfig:
    """Rebind the aggregation entry points of the reference module to the HIP engine.

    After this, ``o1_module.run_majority_vote_inference_experiments(dataset, cache)`` etc. run the
    reference's own driver code with the engine underneath (``batched=False``), or the batched
    drivers above (``batched=True``); plots and results_log_*.json come from the reference's
    unmodified ``helpers.plot_helpers`` either way.
    """
    cfg = config_from_module(o1_module, engine)
    o1_module.process_single_example = lambda example, token_limit, cache, N: process_single_example(
        cfg, example, token_limit, cache, N)
    o1_module.run_experiments = lambda dataset, cache, token_limit, N: run_experiments(
        cfg, dataset, cache, token_limit, N)
    if batched:
        o1_module.run_majority_vote_inference_experiments = (
            lambda dataset, cache, shade_regions=False: run_majority_vote_inference_experiments(
                cfg, dataset, cache, shade_regions))
        o1_module.run_just_ask_nicely_experiments = (
            lambda dataset, cache, run_full_range=False: run_just_ask_nicely_experiments(
                cfg, dataset, cache, run_full_range))
    return cfg


__all__ = [
    "DropInConfig", "config_from_module", "install", "process_single_example", "run_experiments",
    "run_majority_vote_inference_experiments", "run_just_ask_nicely_experiments",
    "majority_vote_budgets", "just_ask_nicely_budgets", "default_engine",
]
