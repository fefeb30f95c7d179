"""Problem-sharded evaluation over the GPUs of one node.

One process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" on CPU for the
tests).  Every (problem, budget) cell is independent (the reference already farms problems out to
independent threads, /root/reference/o1.py:234); only the per-budget reduction o1.py:236-245
crosses problems.  So each rank aggregates its contiguous block of problems into the packed int64
counters and ONE all-reduce(SUM) of counters_size(B) = B*1027 words (65.7 KB at B = 8) finishes the
evaluation.  Integer sums are order-independent => bit-exact at any world size.
"""
from __future__ import annotations


def shard_bounds(P: int, rank: int, world: int):
    """Contiguous block [lo, hi) of problems owned by ``rank`` (SURVEY.md 8e partitioning)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return (rank * P) // world, ((rank + 1) * P) // world


def all_reduce_counters(counters, group=None):
    """In-place SUM all-reduce of the packed per-budget counters (torch tensor, int64)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def all_gather_cells(cells_local, P: int, group=None):
    """Gather the per-cell table [P_local, B, 16] (uint8) of every rank into [P, B, 16]; needed by
    the bootstrap, which resamples over ALL problems."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return cells_local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(P, r, world)[1] - shard_bounds(P, r, world)[0] for r in range(world)]
    pmax = max(sizes)
    pad = torch.zeros((pmax,) + tuple(cells_local.shape[1:]), dtype=cells_local.dtype, device=cells_local.device)
    pad[: cells_local.shape[0]] = cells_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: sizes[r]] for r in range(world)], dim=0)
