"""BASELINE.json config 5: pass@k sweep k in {1, 2, 4, ..., 1024} + 1000-resample bootstrap CI, sharded by
problem over the ranks of one node (SURVEY.md 8a rows a8 / a9 -- NEW semantics, parity unpinned by the
reference, which has neither; the tests check them against the CPU oracle only).

Device pipeline per rank, all on torch's current stream, no host round trip of per-cell data:

  1. vote kernel over the local [P/G, B, N] block   -> cell table (max_count, truth_count, n_modes, hit)
                                                       truth_count is pass@k's c: it comes out of the SAME launch
  2. all_reduce(SUM) of the packed int64 counters   -> accuracy; tie classes present => class bound M
  3. all_gather of the 16-byte cell table           -> every rank holds [P, B] cells (160 KB at P = 10^4, B = 1)
  4. scv_bootstrap for resamples [r R/G, (r+1) R/G) -> int64 [R/G, B, M], LDS-resident code table
  5. (host, once) all_gather of the bootstrap slices; accuracy CI; pass@k sweep from truth_count

With one rank steps 2-3 are no-ops and vote + bootstrap are ONE kernel launch (the workgroups meet at a grid
barrier after their last cell and share the resamples; scv_aggregate_bootstrap_i32), or two launches back to back
when the shape does not allow the fused form.

Errors are collective and cost no host round trip.  The device error word of the engine (an out-of-domain vote makes
the counters invalid; a drawn hit with n_modes >= M makes the resample table invalid) is copied ON DEVICE into one
extra word behind the packed counters (scv_export_error_word) and rides in the counters' all-reduce; ``check`` -- or
``gather_bootstrap``, which exchanges the word once more after the bootstrap launch -- raises on EVERY rank when any
rank saw one.  No rank feeds invalid counters into host floats and none is left waiting in a collective.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import dist as scv_dist
from . import scoring
from ._lib import TIE_CLASSES


@dataclass
class C5Device:
    """Device-side result of one evaluation (nothing here has been synchronised)."""
    counters: object        # int64 [counters_size(B)]  (global after the all-reduce)
    cells: object           # uint8 [P, B, 16]          (all problems, after the all-gather)
    boot: object            # int64 [r1 - r0, B, M]     (this rank's resamples)
    r0: int
    r1: int
    M: int
    flag: object = None     # int64 [1]: sum over the ranks of the device error words after the vote (0 = clean)


def packed_size(B: int) -> int:
    """int64 words of a counters buffer that also carries the collective error word: counters_size(B) + 1."""
    from .engine import counters_size
    return counters_size(B) + 1


def class_bound(counters, B: int) -> int:
    """M = 1 + largest tie class present in the (global) counters.  Reads 66 KB back: a host sync, so callers
    that evaluate repeatedly compute it once (the classes present do not change with the resample seed)."""
    tie = counters[: B * TIE_CLASSES].view(B, TIE_CLASSES)
    present = tie.sum(dim=0).nonzero()
    return int(present.max().item()) + 1 if present.numel() else 1


def evaluate_device(engine, answers_local, truth_local, num_problems: int, resamples: int, seed: int,
                    M: int | None = None, tokens_local=None, n_valid=None, group=None, counters=None,
                    cells_local=None, boot_out=None, fused: bool = True) -> C5Device:
    """Steps 1-4 for this rank.  ``M=None`` derives the class bound from the counters (host sync); with one rank
    and a known M the vote and the bootstrap are one call (``fused=False`` keeps them as two launches).

    ``counters``: int64 [packed_size(B)] = counters | error word (allocated when None; zeroed here).  A buffer of
    counters_size(B) words is accepted too; the error word then travels in a one-word all-reduce of its own.
    Nothing here synchronises the host: call ``check`` (or ``gather_bootstrap(..., engine=engine)``) before trusting
    the numbers."""
    import torch
    import torch.distributed as dist
    from .engine import counters_size
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    B = int(answers_local.shape[1])
    ncount = counters_size(B)
    if counters is None:
        counters = torch.zeros(ncount + 1, dtype=torch.int64, device=answers_local.device)
    else:
        counters.zero_()
    packed = counters.numel() == ncount + 1
    flag = counters[ncount:] if packed else torch.zeros(1, dtype=torch.int64, device=answers_local.device)
    cnt = counters[:ncount]
    export = getattr(engine, "export_error_word", None)
    if world == 1 and M is not None and fused and not scv_dist.collectives_active(group) and hasattr(engine, "aggregate_bootstrap_device"):
        # one rank: the cell table is complete after the vote, so vote + bootstrap go down as ONE call -- one kernel
        # launch when the shape allows it (scv_aggregate_bootstrap_i32)
        _, cells, _, boot = engine.aggregate_bootstrap_device(
            answers_local, truth_local, 0, resamples, seed, M, tokens=tokens_local, n_valid=n_valid, counters=cnt,
            cells=cells_local, out=boot_out)
        if export is not None:
            export(flag)
        return C5Device(cnt, cells, boot, 0, resamples, M, flag)
    _, cells, _ = engine.aggregate_device(answers_local, truth_local, tokens=tokens_local, n_valid=n_valid,
                                          counters=cnt, cells=cells_local)
    if export is not None:
        export(flag)                            # device -> device, on the launch stream: no host round trip
    scv_dist.all_reduce_counters(counters, group)            # counters (+ the error word when packed): ONE all-reduce
    if not packed:
        scv_dist.all_reduce_counters(flag, group)
    all_cells = scv_dist.all_gather_cells(cells, num_problems, group)
    if M is None:
        M = class_bound(cnt, B)
    r0, r1 = (rank * resamples) // world, ((rank + 1) * resamples) // world
    boot = engine.bootstrap_device(all_cells, r0, r1, seed, M, out=boot_out)
    return C5Device(cnt, all_cells, boot, r0, r1, M, flag)


def _raise_collectively(global_word: int, engine, what: str, group=None) -> bool:
    """Every rank raises when the all-reduced error word says some rank's results are invalid: the rank(s) that saw the error
    with the engine's own exception (``engine.sync()`` reads, judges and clears the local word), the others with a remote notice.
    Returns True when the word was non-zero but nothing is wrong: ``engine.sync()`` accepted this rank's word (a fused-barrier
    timeout it has just repaired with a separate bootstrap launch -- bit 2 of scv_export_error_word, one rank only) and there is
    no other rank that could have contributed."""
    from ._lib import ERR_DOMAIN, DomainError, ScvError
    if global_word == 0:
        return False                            # the branch is taken on the EXCHANGED word only: all ranks agree
    local = None
    if engine is not None and hasattr(engine, "sync"):
        try:
            engine.sync()
        except ScvError as e:
            local = e
    if local is not None:
        raise local
    if not scv_dist.collectives_active(group) and engine is not None and hasattr(engine, "sync"):
        return True                             # one rank, and its own sync found nothing to report (it repaired what there was)
    raise DomainError(ERR_DOMAIN, f"another rank reported a device error during {what}; the exchanged results are invalid")


def check(dev: C5Device, engine=None, group=None):
    """Host side of the collective error word after the VOTE (one host sync; every rank takes the same branch because
    the word was summed over the ranks inside the counters' all-reduce).  Call before ``finish_host``."""
    word = int(dev.flag.cpu()[0]) if dev.flag is not None else 0
    if _raise_collectively(word, engine, "the vote", group):
        dev.flag.zero_()                        # judged and cleared by the sync: gather_bootstrap must not count it again


def gather_bootstrap(dev: C5Device, resamples: int, group=None, engine=None):
    """All ranks' resample slices -> int64 [R, B, M] on every rank (step 5's exchange).  With ``engine`` the device
    error word is exchanged once more first (it now also covers the bootstrap launch: a drawn hit with n_modes >= M),
    and every rank raises instead of gathering an invalid table."""
    import torch
    import torch.distributed as dist
    if engine is not None:
        word = torch.zeros(1, dtype=torch.int64, device=dev.boot.device)
        if dev.flag is not None:
            word += dev.flag
        if hasattr(engine, "export_error_word"):
            after = torch.zeros(1, dtype=torch.int64, device=dev.boot.device)
            engine.export_error_word(after)
            word += after
        scv_dist.all_reduce_counters(word, group)
        _raise_collectively(int(word.cpu()[0]), engine, "the vote or the bootstrap", group)
    if not scv_dist.collectives_active(group):
        return dev.boot
    world = dist.get_world_size(group)
    B = int(dev.boot.shape[1])
    sizes = [((r + 1) * resamples) // world - (r * resamples) // world for r in range(world)]
    rmax = max(sizes)                           # all_gather needs equal shapes: pad the shorter slices
    pad = torch.zeros((rmax, B, dev.M), dtype=torch.int64, device=dev.boot.device)
    pad[: dev.boot.shape[0]] = dev.boot
    parts = [torch.empty_like(pad) for _ in range(world)]
    scv_dist.all_gather_tensors(parts, pad, group)
    return torch.cat([parts[r][: sizes[r]] for r in range(world)], dim=0)


def finish_host(counters_np, cells_np, boot_np, num_problems: int, n_valid, ks=scoring.PASS_K_SWEEP):
    """Host floats: accuracy per budget, its bootstrap CI, and the pass@k sweep (one shared float function for
    every path, scoring.pass_at_k_sweep).  cells_np is CELL_DTYPE [P, B]."""
    from .engine import AggregateResult
    P, B = cells_np.shape
    res = AggregateResult.from_counters(counters_np, P, B, cells_np, num_problems=num_problems)
    acc, lo, hi = scoring.bootstrap_percentiles_fast(boot_np, num_problems)
    sweep = scoring.pass_at_k_sweep(np.asarray(n_valid, dtype=np.int64), cells_np["truth_count"], ks)
    return {"accuracy": [res.accuracy(b) for b in range(B)], "ci95": [[float(lo[b]), float(hi[b])] for b in range(B)],
            "bootstrap_accuracy": acc, "pass_at_k": {int(k): [float(x) for x in v] for k, v in sweep.items()}}
