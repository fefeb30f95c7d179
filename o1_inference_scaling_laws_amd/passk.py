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
    and a known M the vote and the bootstrap are one call (``fused=False`` keeps them as two launches)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    B = int(answers_local.shape[1])
    if counters is not None:
        counters.zero_()
    if world == 1 and M is not None and fused and not scv_dist.collectives_active(group) and hasattr(engine, "aggregate_bootstrap_device"):
        # one rank: the cell table is complete after the vote, so vote + bootstrap go down as ONE call -- one kernel
        # launch when the shape allows it (scv_aggregate_bootstrap_i32)
        counters, cells, _, boot = engine.aggregate_bootstrap_device(
            answers_local, truth_local, 0, resamples, seed, M, tokens=tokens_local, n_valid=n_valid, counters=counters,
            cells=cells_local, out=boot_out)
        return C5Device(counters, cells, boot, 0, resamples, M)
    counters, cells, _ = engine.aggregate_device(answers_local, truth_local, tokens=tokens_local, n_valid=n_valid,
                                                 counters=counters, cells=cells_local)
    scv_dist.all_reduce_counters(counters, group)
    all_cells = scv_dist.all_gather_cells(cells, num_problems, group)
    if M is None:
        M = class_bound(counters, B)
    r0, r1 = (rank * resamples) // world, ((rank + 1) * resamples) // world
    boot = engine.bootstrap_device(all_cells, r0, r1, seed, M, out=boot_out)
    return C5Device(counters, all_cells, boot, r0, r1, M)


def gather_bootstrap(dev: C5Device, resamples: int, group=None):
    """All ranks' resample slices -> int64 [R, B, M] on every rank (step 5's exchange)."""
    import torch
    import torch.distributed as dist
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
