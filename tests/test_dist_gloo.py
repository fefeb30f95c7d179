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


class _CpuDeviceEngine:
    """Test double with the engine's DEVICE-mode signature, computing on CPU tensors via the oracle."""

    def aggregate_device(self, answers, truth, tokens=None, n_valid=None, counters=None, cells=None, cell_tokens=None):
        out = coracle.aggregate(answers.numpy(), truth.numpy(), tokens=None if tokens is None else tokens.numpy(),
                                n_valid=None if n_valid is None else n_valid.numpy())
        cnt = torch.from_numpy(_pack(out).copy())
        if counters is not None:
            counters.add_(cnt)                  # DEVICE mode accumulates into the caller's buffer
            cnt = counters
        c = torch.from_numpy(out["cells"].view(np.uint8).reshape(answers.shape[0], answers.shape[1], 16).copy())
        self.rc = out["rc"]
        return cnt, (None if cells is False else c), (torch.from_numpy(out["cell_tokens"]) if tokens is not None else None)

    def aggregate_prefix_device(self, pool, truth, n_valid, tokens=None, counters=None, cells=None, cell_tokens=None):
        # the oracle on the dense expansion answers[p, b, :] = pool[p, :] (what the prefix entry point is defined as)
        B = int(n_valid.shape[0])
        dense = pool.unsqueeze(1).expand(pool.shape[0], B, pool.shape[1]).contiguous()
        dtok = None if tokens is None else tokens.unsqueeze(1).expand(pool.shape[0], B, pool.shape[1]).contiguous()
        return self.aggregate_device(dense, truth, tokens=dtok, n_valid=n_valid, counters=counters, cells=cells)

    def sync(self):
        from o1_inference_scaling_laws_amd._lib import ERR_DOMAIN, DomainError, ScvError
        rc, self.rc = getattr(self, "rc", 0), 0                   # like scv_sync: report and clear
        if rc == ERR_DOMAIN:
            raise DomainError(ERR_DOMAIN, "a vote outside bins 0..1023 was seen; results are invalid")
        if rc != 0:
            raise ScvError(rc, "engine failure (test double)")

    def export_error_word(self, dst):
        """scv_export_error_word: the (device) error word into caller memory, not cleared."""
        dst[0] = 0 if getattr(self, "rc", 0) == 0 else 1

    def bootstrap_device(self, cells, r_begin, r_end, seed, M, out=None):
        c = np.ascontiguousarray(cells.numpy()).view(coracle.CELL_DTYPE).reshape(cells.shape[0], cells.shape[1])
        rc, boot = coracle.bootstrap(c, r_begin, r_end, seed, M)
        if rc != 0:
            self.rc = -2001
        return torch.from_numpy(boot)


def _worker_api(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = scv_dist.shard_bounds(P, rank, world)
        a, t, tr = coracle.synth_fill(hi - lo, B, N, SEED, 1, p_offset=lo, want_tokens=True)
        res = scv_dist.aggregate_sharded(_CpuDeviceEngine(), torch.from_numpy(a), torch.from_numpy(tr), P,
                                         tokens_local=torch.from_numpy(t))
        assert res.cells.shape == (hi - lo, B)
        if rank == 1:
            q.put(([res.accuracy(b) for b in range(B)], [float(res.avg_tokens_used(b)) for b in range(B)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_domain(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from o1_inference_scaling_laws_amd._lib import DomainError
        lo, hi = scv_dist.shard_bounds(P, rank, world)
        a, _, tr = coracle.synth_fill(hi - lo, B, N, SEED, 1, p_offset=lo)
        if rank == 1:
            a[3, 1, 7] = 5000                   # only rank 1 holds an out-of-domain vote
        try:
            scv_dist.aggregate_sharded(_CpuDeviceEngine(), torch.from_numpy(a), torch.from_numpy(tr), P)
            q.put((rank, "no error"))
        except DomainError as e:
            q.put((rank, "local" if "was seen" in str(e) else "remote"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_domain_error_is_collective():
    """a rank with an out-of-domain vote must not return invalid counters, and the other ranks must
    fail with it (the error word rides in the counters' all-reduce), not hang or return silently."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_domain, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: "remote", 1: "local", 2: "remote"}


def test_aggregate_sharded_api_three_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_api, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    acc, avg = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, t, tr = coracle.synth_fill(P, B, N, SEED, 1, want_tokens=True)
    whole = AggregateResult.from_counters(_pack(coracle.aggregate(a, tr, tokens=t)), P, B)
    assert acc == [whole.accuracy(b) for b in range(B)]
    assert avg == [float(whole.avg_tokens_used(b)) for b in range(B)]


PREFIX_NV = [1, 2, 4, 8, 16, 32, 64, 96, 0, 50]


def _worker_prefix(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = scv_dist.shard_bounds(P, rank, world)
        a, t, tr = coracle.synth_fill(hi - lo, 1, N, SEED, 1, p_offset=lo, want_tokens=True)
        nv = torch.tensor(PREFIX_NV, dtype=torch.int32)
        res = scv_dist.aggregate_sharded(_CpuDeviceEngine(), torch.from_numpy(a[:, 0, :].copy()), torch.from_numpy(tr), P,
                                         tokens_local=torch.from_numpy(t[:, 0, :].copy()), n_valid=nv, prefix=True)
        assert res.cells.shape == (hi - lo, len(PREFIX_NV))
        if rank == 0:
            q.put(([res.accuracy(b) for b in range(len(PREFIX_NV))], [float(res.avg_tokens_used(b)) for b in range(len(PREFIX_NV))],
                   res.tie_class_hits.copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_prefix_budgets_sharded_over_two_ranks():
    """The reference's shape (one sample pool per problem, budgets = prefixes: o1.py:274-277) sharded by problem: two
    ranks + one all-reduce reproduce the unsharded dense evaluation bit for bit, floats included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_prefix, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    acc, avg, tie = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, t, tr = coracle.synth_fill(P, 1, N, SEED, 1, want_tokens=True)
    nb = len(PREFIX_NV)
    dense = np.ascontiguousarray(np.broadcast_to(a, (P, nb, N)))
    dtok = np.ascontiguousarray(np.broadcast_to(t, (P, nb, N)))
    whole = AggregateResult.from_counters(_pack(coracle.aggregate(dense, tr, tokens=dtok, n_valid=np.array(PREFIX_NV, dtype=np.int32))), P, nb)
    assert np.array_equal(tie, whole.tie_class_hits)
    assert acc == [whole.accuracy(b) for b in range(nb)]
    assert avg == [float(whole.avg_tokens_used(b)) for b in range(nb)]


def _worker_pipeline(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = scv_dist.CounterPipeline([torch.zeros(counters_size(B), dtype=torch.int64) for _ in range(2)])
        results = []
        for i in range(5):                                    # 5 evaluations over 2 rotating buffers
            buf = pipe.acquire(i)
            lo, hi = scv_dist.shard_bounds(P, rank, world)
            a, t, tr = coracle.synth_fill(hi - lo, B, N, SEED + i, 1, p_offset=lo, want_tokens=True)
            buf += torch.from_numpy(_pack(coracle.aggregate(a, tr, tokens=t)))
            pipe.publish(i)
            if i >= 1:                                        # read evaluation i-1 only after its reduce finished
                pipe.pending[(i - 1) % 2] and pipe.pending[(i - 1) % 2].wait()
                results.append(pipe.buffers[(i - 1) % 2].clone())
        pipe.drain()
        results.append(pipe.buffers[4 % 2].clone())
        if rank == 0:
            q.put([r.numpy() for r in results])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_counter_pipeline_overlapped_all_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(results) == 5
    for i, got in enumerate(results):
        a, t, tr = coracle.synth_fill(P, B, N, SEED + i, 1, want_tokens=True)
        assert np.array_equal(got, _pack(coracle.aggregate(a, tr, tokens=t))), i
    single = scv_dist.CounterPipeline([torch.zeros(4, dtype=torch.int64)])
    single.acquire(0).add_(3)
    assert single.publish(0).tolist() == [3, 3, 3, 3] and single.acquire(1).tolist() == [0, 0, 0, 0]


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


def _worker_c5_error(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from o1_inference_scaling_laws_amd import passk
        from o1_inference_scaling_laws_amd._lib import DomainError, ScvError
        lo, hi = scv_dist.shard_bounds(P, rank, world)
        a, _, tr = coracle.synth_fill(hi - lo, 1, N, SEED, 3, p_offset=lo)
        if mode == "domain" and rank == 1:
            a[2, 0, 5] = 7777                    # only rank 1 holds an out-of-domain vote
        eng = _CpuDeviceEngine()
        M = 1 if mode == "overflow" else None    # D3 has 2-/3-way ties: M = 1 overflows in the bootstrap (on every rank that draws one)
        try:
            d = passk.evaluate_device(eng, torch.from_numpy(a), torch.from_numpy(tr), P, 40, 9, M=M)
            passk.check(d, eng)                  # after the vote
            boot = passk.gather_bootstrap(d, 40, engine=eng)
            q.put((rank, "ok", tuple(boot.shape)))
        except DomainError as e:
            q.put((rank, "domain-local" if "was seen" in str(e) else "domain-remote", None))
        except ScvError as e:
            q.put((rank, "scv", None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_c5(mode, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c5_error, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, what, shape = q.get(timeout=180)
        got[r] = (what, shape)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_c5_pipeline_errors_are_collective():
    """passk.evaluate_device on two ranks.  One bad vote on rank 1 makes EVERY rank raise in
    passk.check (the error word rode in the counters' all-reduce: rank 1 with its own error, rank 0 with the remote
    notice) instead of feeding invalid counters into accuracy / CI / pass@k while a peer sits in gather_bootstrap; a
    class bound that overflows in the bootstrap is caught by gather_bootstrap's exchange of the word on every rank;
    a clean run returns the whole table on both."""
    assert _run_c5("clean") == {0: ("ok", (40, 1, 4)), 1: ("ok", (40, 1, 4))}
    assert _run_c5("domain") == {0: ("domain-remote", None), 1: ("domain-local", None)}
    got = _run_c5("overflow")
    assert {v[0] for v in got.values()} <= {"scv", "domain-remote"} and len(got) == 2 and "ok" not in {v[0] for v in got.values()}


# ---- rehearsal at the target rank count (north_star: "shard across the 8 GPUs of one node"): world = 8, P = 10 000 ----------

P8, B8, N8, R8 = 10_000, 2, 24, 1000


def _worker_eight(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from o1_inference_scaling_laws_amd import passk
        lo, hi = scv_dist.shard_bounds(P8, rank, world)
        assert hi - lo == 1250
        a, t, tr = coracle.synth_fill(hi - lo, B8, N8, SEED, 3, p_offset=lo, want_tokens=True)
        eng = _CpuDeviceEngine()
        # C4's shape: block of 1250 problems -> packed counters -> ONE all-reduce
        res = scv_dist.aggregate_sharded(eng, torch.from_numpy(a), torch.from_numpy(tr), P8, tokens_local=torch.from_numpy(t))
        # C5's shape on budget 0: vote -> all-reduce (+ error word) -> cell all-gather -> 125 resamples per rank -> gather
        a0 = np.ascontiguousarray(a[:, :1, :])
        d = passk.evaluate_device(eng, torch.from_numpy(a0), torch.from_numpy(tr), P8, R8, 77)
        assert (d.r1 - d.r0) == R8 // world and tuple(d.cells.shape) == (P8, 1, 16)
        passk.check(d, eng)
        boot = passk.gather_bootstrap(d, R8, engine=eng)
        q.put((rank, res.tie_class_hits.copy(), res.token_sum.copy(), res.truth_count_sum.copy(), [res.accuracy(b) for b in range(B8)],
               d.counters.numpy().copy(), d.cells.numpy().copy(), boot.numpy().copy(), d.M))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_eight_ranks_of_1250_problems_equal_the_unsharded_evaluation():
    """VERDICT r4 next #1b: world = 8 with P = 10 000 (1250-problem shards; C5: 1000 resamples = 125 per rank) on gloo.
    EVERY rank's all-reduced counters, gathered cell table and gathered resample table equal the unsharded oracle run."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_eight, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=600)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, t, tr = coracle.synth_fill(P8, B8, N8, SEED, 3, want_tokens=True)
    whole = coracle.aggregate(a, tr, tokens=t)
    whole0 = coracle.aggregate(np.ascontiguousarray(a[:, :1, :]), tr)
    M = got[0][7]
    rc, want_boot = coracle.bootstrap(whole0["cells"], 0, R8, 77, M)
    assert rc == 0 and M >= 3                      # D3: two- and three-way ties are present
    ref_acc = [AggregateResult.from_counters(_pack(whole), P8, B8).accuracy(b) for b in range(B8)]
    for r in range(world):
        tie, tok, tcs, acc, c5_counters, c5_cells, boot, m = got[r]
        assert np.array_equal(tie, whole["tie_class_hits"]) and np.array_equal(tok, whole["token_sum"]), r
        assert np.array_equal(tcs, whole["truth_count_sum"]) and acc == ref_acc, r
        assert m == M and np.array_equal(c5_counters, _pack(whole0)), r
        assert np.array_equal(c5_cells, whole0["cells"].view(np.uint8).reshape(P8, 1, 16)), r
        assert boot.shape == (R8, 1, M) and np.array_equal(boot, want_boot), r
