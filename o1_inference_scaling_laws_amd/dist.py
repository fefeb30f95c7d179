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


def collectives_active(group=None) -> bool:
    """True when the exchange steps must really run: a process group with more than one rank -- or with ONE rank when
    SCV_FORCE_COLLECTIVES=1, which is how a 1-GPU box exercises the RCCL calls themselves (int64 all-reduce, uint8 /
    int64 all-gathers, barrier) under the same code path an 8-GPU node takes."""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("SCV_FORCE_COLLECTIVES") == "1"


def all_reduce_counters(counters, group=None):
    """In-place SUM all-reduce of the packed per-budget counters (torch tensor, int64)."""
    import torch.distributed as dist
    if collectives_active(group):
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def all_gather_tensors(parts, tensor, group=None):
    """dist.all_gather, except that the gloo backend (CPU tests, and the 1-GPU box's shared-device runs) cannot
    gather CUDA tensors: there the exchange is staged through host copies.  RCCL gathers in place over xGMI."""
    import torch.distributed as dist
    if tensor.is_cuda and dist.get_backend(group) == "gloo":
        host_parts = [p.cpu() for p in parts]
        dist.all_gather(host_parts, tensor.cpu(), group=group)
        for p, h in zip(parts, host_parts):
            p.copy_(h)
        return
    dist.all_gather(parts, tensor, group=group)


def all_gather_cells(cells_local, P: int, group=None):
    """Gather the per-cell table [P_local, B, 16] (uint8) of every rank into [P, B, 16]; needed by
    the bootstrap, which resamples over ALL problems."""
    import torch
    import torch.distributed as dist
    if not collectives_active(group):
        return cells_local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(P, r, world)[1] - shard_bounds(P, r, world)[0] for r in range(world)]
    pmax = max(sizes)
    pad = torch.zeros((pmax,) + tuple(cells_local.shape[1:]), dtype=cells_local.dtype, device=cells_local.device)
    pad[: cells_local.shape[0]] = cells_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    all_gather_tensors(parts, pad, group)
    return torch.cat([parts[r][: sizes[r]] for r in range(world)], dim=0)


def aggregate_sharded(engine, answers_local, truth_local, num_problems: int, tokens_local=None, n_valid=None,
                      group=None, want_cells=True, prefix=False):
    """One sharded evaluation: this rank's contiguous block of problems goes through
    ``engine.aggregate_device`` (any object with that method: the HIP engine in production), the packed
    int64 counters are summed over the ranks with ONE all-reduce, and the reference's floats are taken
    over the GLOBAL number of problems.  Returns an ``AggregateResult`` whose counters are global and
    whose cell table is this rank's block.

    Errors are collective: whatever ``engine.sync()`` raises on a rank (SCV_ERR_DOMAIN: a vote outside bins
    0..1023, which makes the counters invalid; a HIP error; any other ``ScvError``) is held back, its code rides in
    one extra word of the same all-reduce, and EVERY rank raises when any rank saw one -- the rank itself with its own
    exception, the others with a ``DomainError`` / ``ScvError`` that says so.  No rank returns counters the ABI
    documents as invalid, and no rank is left waiting in a collective the others skipped.

    ``prefix=True``: ``answers_local`` / ``tokens_local`` are this rank's block of ONE sample pool per problem,
    ``[P_local, N]``, and budget b votes over its first ``n_valid[b]`` samples (the reference's shape, o1.py:274-277);
    the block goes through ``engine.aggregate_prefix_device`` and everything else is the same.  ``n_valid`` may then be a
    HOST sequence (list / numpy array) instead of a device tensor: the budgets are known at the call, and a list of the served
    form (powers of two over pools of 17 .. 128 votes) is promised to the library -- one launch, no decision on the device."""
    import torch
    from ._lib import ERR_DOMAIN, DomainError, ScvError
    from .engine import AggregateResult, cells_from_torch, counters_size
    budgets_host = None
    if prefix:
        if n_valid is None:
            raise ValueError("prefix=True needs n_valid[B]")
        if not hasattr(n_valid, "is_cuda"):
            # the budgets as a HOST sequence (list / numpy): the rank knows them, so the engine may promise them to the library
            # (one launch instead of two deciding from device memory; Engine.aggregate_prefix_device) -- and puts them on the device
            budgets_host = [int(x) for x in n_valid]
            n_valid = torch.tensor(budgets_host, dtype=torch.int32, device=answers_local.device)
        P_local, B = int(answers_local.shape[0]), int(n_valid.shape[0])
    else:
        P_local, B = int(answers_local.shape[0]), int(answers_local.shape[1])
    ncount = counters_size(B)
    packed = torch.zeros(ncount + 1, dtype=torch.int64, device=answers_local.device)     # counters | error word
    if prefix:
        kw = {"budgets_host": budgets_host} if budgets_host is not None else {}
        _, cells, cell_tokens = engine.aggregate_prefix_device(
            answers_local, truth_local, n_valid, tokens=tokens_local, counters=packed[:ncount],
            cells=None if want_cells else False, **kw)
    else:
        _, cells, cell_tokens = engine.aggregate_device(
            answers_local, truth_local, tokens=tokens_local, n_valid=n_valid, counters=packed[:ncount],
            cells=None if want_cells else False)
    local_error = None
    if hasattr(engine, "sync"):
        try:
            engine.sync()                       # DEVICE mode reports its errors here and only here
        except ScvError as e:                   # DomainError, HIP errors, barrier / bootstrap errors: all collective
            local_error = e
            packed[ncount] = 1 if isinstance(e, DomainError) else (1 << 20)
    all_reduce_counters(packed, group)
    host = packed.cpu().numpy()
    if host[ncount] != 0:
        if local_error is not None:
            raise local_error
        if int(host[ncount]) >> 20:
            raise ScvError(int(-2001), "another rank failed in the engine; the all-reduced counters are invalid")
        raise DomainError(ERR_DOMAIN, "another rank saw a vote outside bins 0..1023; the all-reduced counters are invalid")
    host_cells = cells_from_torch(cells) if cells is not None else None
    host_ctok = cell_tokens.cpu().numpy() if cell_tokens is not None else None
    return AggregateResult.from_counters(host[:ncount], P_local, B, host_cells, host_ctok, num_problems=num_problems)


class CounterPipeline:
    """Rotating packed-counter buffers so the (latency-bound, 65.7 KB) all-reduce of evaluation i runs on
    RCCL's own stream while the kernel of evaluation i+1 streams its chunk: ``acquire(i)`` waits for the
    buffer's previous all-reduce and zeroes it, ``publish(i)`` starts the asynchronous all-reduce,
    ``drain()`` waits for everything outstanding.  With one rank it degenerates to plain buffers."""

    def __init__(self, buffers, group=None):
        self.buffers = list(buffers)
        self.pending = [None] * len(self.buffers)
        self.group = group

    def _distributed(self):
        return collectives_active(self.group)

    def acquire(self, i: int, zero: bool = True):
        """``zero=False`` for evaluations that overwrite their counters (Engine.aggregate_device(overwrite=True))."""
        k = i % len(self.buffers)
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        if zero:
            self.buffers[k].zero_()
        return self.buffers[k]

    def publish(self, i: int):
        import torch.distributed as dist
        k = i % len(self.buffers)
        if self._distributed():
            self.pending[k] = dist.all_reduce(self.buffers[k], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return self.buffers[k]

    def drain(self):
        for k, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[k] = None
