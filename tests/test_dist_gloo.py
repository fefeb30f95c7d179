"""N > 1 path on CPU: world_size 2, gloo, 127.0.0.1.  Each rank aggregates its contiguous block of
problems (oracle adapter standing in for the GPU) and ONE all-reduce of the packed int64 counters
reproduces the unsharded evaluation bit for bit."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from o1_inference_scaling_laws_amd import dist as scv_dist
from o1_inference_scaling_laws_amd.engine import AggregateResult, counters_size
from oracle import coracle

P, B, N, SEED, DIST = 31, 3, 96, 4242, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pack(out):
    return np.concatenate([out["tie_class_hits"].reshape(-1), out["token_sum"], out["truth_count_sum"]])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = scv_dist.shard_bounds(P, rank, world)
        a, t, tr = coracle.synth_fill(hi - lo, B, N, SEED, DIST, p_offset=lo, want_tokens=True)
        nv = np.array([N, N // 2, 1], dtype=np.int32)
        out = coracle.aggregate(a, tr, tokens=t, n_valid=nv)
        counters = torch.from_numpy(_pack(out).copy())
        assert counters.numel() == counters_size(B)
        scv_dist.all_reduce_counters(counters)
        cells = torch.from_numpy(out["cells"].view(np.uint8).reshape(hi - lo, B, 16).copy())
        gathered = scv_dist.all_gather_cells(cells, P)
        if rank == 0:
            q.put((counters.numpy().copy(), gathered.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_all_reduce_equals_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    counters, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, t, tr = coracle.synth_fill(P, B, N, SEED, DIST, want_tokens=True)
    nv = np.array([N, N // 2, 1], dtype=np.int32)
    whole = coracle.aggregate(a, tr, tokens=t, n_valid=nv)
    assert np.array_equal(counters, _pack(whole))
    assert np.array_equal(gathered, whole["cells"].view(np.uint8).reshape(P, B, 16))
    res = AggregateResult.from_counters(counters, P, B)
    assert res.accuracy(0) == AggregateResult.from_counters(_pack(whole), P, B).accuracy(0)
    assert res.tie_class_hits[:, 2:4].sum() > 0      # the tie classes really were exercised
