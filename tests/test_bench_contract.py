"""bench.py contract on the GPU box: one JSON line with the driver's keys (+ roofline, cpu_baseline), and the
N > 1 control flow (two ranks, gloo, both on cuda:0 -- everything except RCCL itself) checked BIT FOR BIT against
single-rank runs over the same global problem indices: the C3/C4 path (sharded chunks + one all-reduce of the
packed counters) and the C5 path (cell all-gather + resample split)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def _bench(extra, ranks=1, timeout=900, self_spawn=False, env=None):
    if ranks == 1 and not self_spawn:
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), *extra]
    elif self_spawn:
        # the driver's own form: plain `python bench.py --gpus N ...`, no launcher -- bench.py starts its ranks itself
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(ranks), *extra]
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(ranks), *extra,
               "--backend", "gloo", "--share-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return _last_json(out.stdout)


def test_single_gpu_line_has_the_contract_fields():
    d = _bench(["--problems-per-step", "48", "--samples", str(1 << 16), "--steps", "3", "--warmup", "1", "--cpu-baseline-seconds", "0.5"])
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "int32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["hip_devices_visible"] >= 1 and d["config"]["rccl_ranks"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert r["traffic"] is None and r["traffic_measured_in_this_run"] is None     # not the default workload: nothing injected
    c = d["cpu_baseline"]
    # value = the UNMODIFIED reference loop (o1.run_experiments from oracle/_ref), timed on this box's host
    assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] > 0 and "sample" in c and "o1.run_experiments" in c["what"]
    assert [r["N"] for r in c["reference_loop"]] == [256, 2048, 8192] and all(r["accuracy_matches_restatement"] for r in c["reference_loop"])
    a = c["arithmetic"]
    assert a["kind"] == "port" and "statistics.multimode" in a["what"] and a["value"] > c["value"]   # without pools and key lookups
    assert a["all_cores"]["value"] > 0 and c["c_port"]["value"] > a["value"]      # C port beats the Python loop
    assert d["config"]["chunks_distinct"] == d["config"]["resident_chunks"] == 4  # min(8 chunks of C3, steps + warmup)
    assert d["parity"].startswith("bit-exact") and "statistics.multimode" in d["parity"] and "bit-exact" in d["metric"]
    assert abs(d["value"] - 48 * 8 * (1 << 16) * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9


def test_two_ranks_equal_one_rank_word_for_word(tmp_path):
    """Weak scaling: step s of a G-rank run covers global problems [s G Pc, (s+1) G Pc).  Two ranks x 48 problems must
    give the SAME all-reduced counters (8216 int64 words) as one rank x 96 problems."""
    common = ["--samples", str(1 << 15), "--steps", "3", "--warmup", "1", "--resident", "4"]
    two = _bench(["--problems-per-step", "48", *common, "--dump", str(tmp_path / "two.npz")], ranks=2)
    one = _bench(["--problems-per-step", "96", *common, "--dump", str(tmp_path / "one.npz"), "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["cpu_baseline"] is None and two["parity"].startswith("bit-exact: EVERY rank's first 32 problems") and two["allreduce_verified"] is True and two["parity_ranks_checked"] == 2
    assert two["config"]["backend"] == "gloo" and two["config"]["devices_shared_by_ranks"] is True
    a, b = np.load(tmp_path / "two.npz")["counters"], np.load(tmp_path / "one.npz")["counters"]
    assert a.shape == b.shape == (8 * 1027,) and np.array_equal(a, b) and a.sum() > 0
    assert two["accuracy_last_step"] == one["accuracy_last_step"]


def test_c5_two_ranks_equal_one_rank_and_oracle(tmp_path):
    """C5 (vote -> counters all-reduce -> cell all-gather -> R/G resamples per rank -> gather): counters, the
    gathered cell table and the whole bootstrap table of a 2-rank run equal the 1-rank run's, which bench.py
    itself compares with the oracle (cells + all resample counters)."""
    common = ["--workload", "c5", "--problems", "300", "--samples", str(1 << 14), "--resamples", "101", "--steps", "2", "--warmup", "1", "--dist", "3"]
    one = _bench([*common, "--dump", str(tmp_path / "one.npz"), "--cpu-baseline-seconds", "0.3"])
    two = _bench([*common, "--dump", str(tmp_path / "two.npz"), "--no-cpu-baseline"], ranks=2)
    assert one["scaling"] == "strong" and "bootstrap_parity" in one["c5"] and one["parity"].startswith("bit-exact")
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    for k in ("counters", "cells", "boot"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    assert a["boot"].shape[0] == 101 and a["cells"].shape[:2] == (300, 1)
    assert one["c5"]["pass_at_k"] == two["c5"]["pass_at_k"] and one["c5"]["accuracy_ci95"] == two["c5"]["accuracy_ci95"]
    assert abs(one["value"] - 300 * (1 << 14) * 2 / (one["ms_per_step"] * 2e-3)) / one["value"] < 1e-9


def test_rccl_calls_execute_on_one_rank(tmp_path):
    """The gpurun boxes have one GPU, so RCCL across GPUs cannot run here; what CAN run is every RCCL call of the
    N > 1 path on a one-rank NCCL process group (SCV_FORCE_COLLECTIVES=1): the async int64 all-reduce of the packed
    counters behind each step, the MAX all-reduces of the timing, the barrier, and for C5 the uint8 all-gather of the
    cell table and the int64 all-gather of the resample slices.  Results must equal the plain single-process run."""
    env = dict(os.environ, SCV_FORCE_COLLECTIVES="1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--backend", "nccl"]
    c3 = ["--problems-per-step", "48", "--samples", str(1 << 15), "--steps", "3", "--warmup", "1", "--resident", "4"]
    out = subprocess.run([*base, *c3, "--dump", str(tmp_path / "rccl.npz")], capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["collectives_forced_on_one_rank"] is True
    plain = _bench([*c3, "--dump", str(tmp_path / "plain.npz"), "--no-cpu-baseline"])
    assert np.array_equal(np.load(tmp_path / "rccl.npz")["counters"], np.load(tmp_path / "plain.npz")["counters"])
    assert d["accuracy_last_step"] == plain["accuracy_last_step"]
    c5 = ["--workload", "c5", "--problems", "300", "--samples", str(1 << 14), "--resamples", "101", "--steps", "2", "--warmup", "1", "--dist", "3"]
    out = subprocess.run([*base, *c5, "--dump", str(tmp_path / "rccl5.npz")], capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    plain5 = _bench([*c5, "--dump", str(tmp_path / "plain5.npz"), "--no-cpu-baseline"])
    a, b = np.load(tmp_path / "rccl5.npz"), np.load(tmp_path / "plain5.npz")
    for k in ("counters", "cells", "boot"):
        assert np.array_equal(a[k], b[k]), k
    assert _last_json(out.stdout)["c5"]["pass_at_k"] == plain5["c5"]["pass_at_k"]


def test_c2_steps_captured_in_one_graph_launch():
    """C2 (30 x 8 x 2^17, one kernel launch per evaluation): --graph --graph-steps K replays K evaluations per graph
    launch; steps round up to a multiple of K, the line keeps the contract and the cells of the last timed step are
    checked against the oracle inside bench.py."""
    d = _bench(["--workload", "c2", "--steps", "10", "--warmup", "3", "--graph", "--graph-steps", "4", "--cpu-baseline-seconds", "0.2"])
    assert KEYS <= set(d) and d["steps"] == 12 and d["warmup"] == 4
    assert d["config"]["launch"] == "hipGraph replay (4 step(s) per graph launch)"
    assert d["parity"].startswith("bit-exact: 30 problems x 8 budgets x 131072 votes of the last TIMED chunk")
    assert abs(d["value"] - 30 * 8 * (1 << 17) * 12 / (d["ms_per_step"] * 12e-3)) / d["value"] < 1e-9


def test_plain_python_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """The driver's multi-GPU form is the single-GPU command with another number: `python bench.py --gpus N`, no
    torch.distributed.run in front.  bench.py then starts N ranks itself (the reference's run_experiments fans out
    itself too, o1.py:232-240) and relays exactly one JSON line.  Two self-started ranks (gloo, sharing cuda:0) must
    give the 1-rank run's counters word for word."""
    common = ["--samples", str(1 << 15), "--steps", "3", "--warmup", "1", "--resident", "4"]
    two = _bench(["--problems-per-step", "48", *common, "--backend", "gloo", "--share-device", "--dump", str(tmp_path / "two.npz")],
                 ranks=2, self_spawn=True)
    one = _bench(["--problems-per-step", "96", *common, "--dump", str(tmp_path / "one.npz"), "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["config"]["ranks_started_by"].startswith("bench.py itself")
    assert two["config"]["backend"] == "gloo" and two["config"]["rccl_ranks"] is None
    a, b = np.load(tmp_path / "two.npz")["counters"], np.load(tmp_path / "one.npz")["counters"]
    assert a.shape == (8 * 1027,) and np.array_equal(a, b) and a.sum() > 0
    assert two["accuracy_last_step"] == one["accuracy_last_step"]


def test_plain_python_bench_runs_rccl_on_a_self_started_rank(tmp_path):
    """Same entry, backend nccl: with SCV_FORCE_COLLECTIVES=1 the self-started (single) rank builds a real RCCL
    communicator and every collective of the N > 1 path executes; rccl_ranks equals the world size (asserted inside
    bench.py too) and the counters equal the plain run's."""
    c3 = ["--problems-per-step", "48", "--samples", str(1 << 15), "--steps", "3", "--warmup", "1", "--resident", "4", "--no-cpu-baseline"]
    d = _bench([*c3, "--backend", "nccl", "--dump", str(tmp_path / "rccl.npz")], ranks=1, self_spawn=True,
               env=dict(os.environ, SCV_FORCE_COLLECTIVES="1"))
    assert d["config"]["rccl_ranks"] == 1 == d["n_gpus"] and d["config"]["collectives_forced_on_one_rank"] is True
    assert d["config"]["ranks_started_by"].startswith("bench.py itself")
    assert d["config"]["hsa_enable_ipc_mode_legacy"]["source"]                   # inherited, probed, or left unset: always said
    plain = _bench([*c3, "--dump", str(tmp_path / "plain.npz")])
    assert np.array_equal(np.load(tmp_path / "rccl.npz")["counters"], np.load(tmp_path / "plain.npz")["counters"])


def test_more_ranks_than_gpus_is_one_json_error_line():
    """`python bench.py --gpus 8` on a 1-GPU box: exit status != 0 and ONE JSON line that says why."""
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode != 0
    d = _last_json(out.stdout)
    assert "error" in d and f"--gpus {n}" in d["error"] and d["n_gpus"] == n and d["value"] is None


def test_single_process_communicator_runs_equal_the_torch_run_word_for_word(tmp_path):
    """one bench, both multi-GPU stacks.  `--comm peer --gpus 2 --share-device` (ONE process, the library's
    one-shot all-reduce over peer access; two contexts on cuda:0) and `--comm rccl` on one device (single-process RCCL: the
    ncclAllReduce call path itself) give the torch run's all-reduced counters word for word, with the same JSON contract plus
    per-rank kernel times and the exposed all-reduce time; the create-time self-test ran."""
    common = ["--samples", str(1 << 15), "--steps", "3", "--warmup", "1", "--resident", "4", "--no-cpu-baseline"]
    one = _bench(["--problems-per-step", "96", *common, "--dump", str(tmp_path / "one.npz")])
    peer = _bench(["--comm", "peer", "--gpus", "2", "--share-device", "--problems-per-step", "48", *common, "--dump", str(tmp_path / "peer.npz")])
    a, b = np.load(tmp_path / "peer.npz")["counters"], np.load(tmp_path / "one.npz")["counters"]
    assert a.shape == b.shape == (8 * 1027,) and np.array_equal(a, b) and a.sum() > 0
    assert KEYS <= set(peer) and peer["n_gpus"] == 2 and peer["config"]["comm"] == "peer" and one["config"]["comm"] == "torch"
    assert peer["config"]["comm_ranks"] == 2 and peer["config"]["comm_selftest_words_per_rank"] > 0 and peer["config"]["devices"] == [0, 0]
    r = peer["roofline"]
    assert r["kernel_avg_ms_per_rank_min"] <= r["kernel_avg_ms_per_rank_max"] == r["kernel_avg_ms"] and r["exposed_allreduce_us"] > 0
    assert peer["accuracy_last_step"] == one["accuracy_last_step"]
    assert abs(peer["value"] - 2 * 48 * 8 * (1 << 15) * 3 / (peer["ms_per_step"] * 3e-3)) / peer["value"] < 1e-9
    rccl = _bench(["--comm", "rccl", "--gpus", "1", "--problems-per-step", "96", *common, "--dump", str(tmp_path / "rccl.npz")])
    assert np.array_equal(np.load(tmp_path / "rccl.npz")["counters"], b)
    assert rccl["config"]["comm"] == "rccl" and rccl["config"]["rccl_ranks"] == 1 and rccl["config"]["comm_selftest_words_per_rank"] > 0
    # parity leg of the single-process line, and C5 through the communicator (cells + whole resample table vs the oracle inside bench.py)
    c5 = ["--workload", "c5", "--problems", "300", "--samples", str(1 << 14), "--resamples", "101", "--steps", "2", "--warmup", "1", "--dist", "3"]
    p5 = _bench(["--comm", "peer", "--gpus", "3", "--share-device", *c5, "--dump", str(tmp_path / "p5.npz")])
    t5 = _bench([*c5, "--dump", str(tmp_path / "t5.npz"), "--no-cpu-baseline"])
    assert p5["parity"].startswith("bit-exact: EVERY rank's first 32 problems") and p5["allreduce_verified"] is True and p5["parity_ranks_checked"] == 3 and "bootstrap_parity" in p5["c5"] and p5["c5"]["device_error_word"] == 0
    x, y = np.load(tmp_path / "p5.npz"), np.load(tmp_path / "t5.npz")
    for k in ("counters", "cells", "boot"):
        assert x[k].shape == y[k].shape and np.array_equal(x[k], y[k]), k


def test_single_process_communicator_with_too_few_gpus_is_one_json_error_line():
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--comm", "peer", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode != 0
    d = _last_json(out.stdout)
    assert "error" in d and f"--gpus {n}" in d["error"] and d["value"] is None


# ---- rehearsal at the target rank count: 8 ranks, P = 10 000 in shards of 1250, every stack, C3/C4 and C5 ----

EIGHT = ["--samples", "256", "--steps", "2", "--warmup", "1", "--resident", "2"]


def _assert_self_verified(d, ranks, oracle=True):
    x = d["exchange_check"]
    assert d["allreduce_verified"] is True and x["allreduce_verified"] is True and x["ranks_verified"] == ranks
    assert x["allreduce_words"] == (8 * 1027 if "C5" not in d["config"]["workload"] else 1027)
    if oracle:
        assert d["parity_ranks_checked"] == ranks and x["parity_problems_per_rank"] == 32
        assert d["parity"].startswith("bit-exact: EVERY rank's first 32 problems")
    else:
        assert d["parity_ranks_checked"] == 0


def test_eight_ranks_c4_shape_every_stack_equals_one_rank(tmp_path):
    """C4's shape at the target rank count on the 1-GPU box: 8 ranks x 1250 problems per step (a 10 000-problem step), torch/gloo
    processes and ONE process over the library's peer communicator (8 contexts on cuda:0), and single-process RCCL on one device,
    all equal to one rank x 10 000 problems word for word -- and each multi-rank line has verified its own exchange step on EVERY
    rank (pre-reduce counters gathered independently, numpy sum == every rank's all-reduced buffer) and sent 32 problems of EVERY
    rank's last chunk through the oracle."""
    one = _bench(["--problems-per-step", "10000", *EIGHT, "--no-cpu-baseline", "--dump", str(tmp_path / "one.npz")])
    want = np.load(tmp_path / "one.npz")["counters"]
    assert want.shape == (8 * 1027,) and want.sum() > 0
    tor = _bench(["--problems-per-step", "1250", *EIGHT, "--dump", str(tmp_path / "torch8.npz")], ranks=8)
    assert tor["n_gpus"] == 8 and tor["config"]["backend"] == "gloo" and np.array_equal(np.load(tmp_path / "torch8.npz")["counters"], want)
    _assert_self_verified(tor, 8)
    peer = _bench(["--comm", "peer", "--gpus", "8", "--share-device", "--problems-per-step", "1250", *EIGHT, "--dump", str(tmp_path / "peer8.npz")])
    assert peer["n_gpus"] == 8 and peer["config"]["devices"] == [0] * 8 and np.array_equal(np.load(tmp_path / "peer8.npz")["counters"], want)
    _assert_self_verified(peer, 8)
    rccl = _bench(["--comm", "rccl", "--gpus", "1", "--problems-per-step", "10000", *EIGHT, "--dump", str(tmp_path / "rccl1.npz")])
    assert np.array_equal(np.load(tmp_path / "rccl1.npz")["counters"], want) and rccl["config"]["rccl_ranks"] == 1
    _assert_self_verified(rccl, 1)
    assert tor["accuracy_last_step"] == peer["accuracy_last_step"] == rccl["accuracy_last_step"] == one["accuracy_last_step"]


def test_eight_ranks_c5_every_stack_equals_one_rank(tmp_path):
    """C5 at the target rank count: P = 10 000 in shards of 1250, R = 1000 resamples = 125 per rank; counters, the gathered
    cell table and the whole resample table of the 8-rank runs (torch/gloo processes; one process, peer communicator) equal
    the 1-rank run's, which bench.py compares with the oracle; every rank verified its gathered tables."""
    c5 = ["--workload", "c5", "--problems", "10000", "--samples", "256", "--resamples", "1000", "--steps", "2", "--warmup", "1", "--dist", "3"]
    one = _bench([*c5, "--dump", str(tmp_path / "one.npz"), "--cpu-baseline-seconds", "0.3"])
    assert "bootstrap_parity" in one["c5"]
    tor = _bench([*c5, "--dump", str(tmp_path / "t8.npz")], ranks=8)
    peer = _bench(["--comm", "peer", "--gpus", "8", "--share-device", *c5, "--dump", str(tmp_path / "p8.npz")])
    a = np.load(tmp_path / "one.npz")
    for name in ("t8", "p8"):
        b = np.load(tmp_path / f"{name}.npz")
        for k in ("counters", "cells", "boot"):
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (name, k)
    assert a["boot"].shape[0] == 1000 and a["cells"].shape[:2] == (10000, 1)
    _assert_self_verified(tor, 8)
    _assert_self_verified(peer, 8)
    assert tor["exchange_check"]["c5_resample_table_equal_on_every_rank"] is True
    assert "bootstrap_parity" in tor["c5"] and "bootstrap_parity" in peer["c5"]
    assert one["c5"]["pass_at_k"] == tor["c5"]["pass_at_k"]


def test_multi_rank_line_checks_the_closed_form_of_the_degenerate_distribution():
    """--dist 2: every cell is a strict win, so tie_class_hits[b][1] == ranks x problems and truth_count_sum[b] == that x N whatever
    the sharding; the multi-rank lines check it on the independently gathered sum."""
    d = _bench(["--problems-per-step", "100", "--samples", "512", "--steps", "2", "--warmup", "1", "--resident", "2", "--dist", "2"], ranks=4)
    assert d["exchange_check"]["closed_form"].startswith("D2: tie_class_hits[b][1] == 400") and d["allreduce_verified"] is True
    p = _bench(["--comm", "peer", "--gpus", "4", "--share-device", "--problems-per-step", "100", "--samples", "512", "--steps", "2", "--warmup", "1",
                "--resident", "2", "--dist", "5"])
    assert p["exchange_check"]["closed_form"].startswith("D5: tie_class_hits[b][1] == 0") and p["allreduce_verified"] is True


@pytest.mark.parametrize("stack", ["torch", "peer"])
def test_a_wrong_sum_on_one_rank_stops_the_line(stack):
    """The check has teeth: one word of ONE rank's all-reduced buffer off by one (what a stale peer read would look like) => no
    bench line, exit status != 0, and the error names the rank and the word."""
    env = dict(os.environ, SCV_BENCH_FAULT="2:1030")
    base = ["--problems-per-step", "64", "--samples", "256", "--steps", "2", "--warmup", "1", "--resident", "2", "--no-cpu-baseline"]
    if stack == "torch":
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "3", *base, "--backend", "gloo", "--share-device"]
    else:
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--comm", "peer", "--gpus", "3", "--share-device", *base]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert out.returncode != 0
    text = out.stdout + out.stderr
    assert "rank 2: all-reduced counters != independently gathered sum: word 1030" in text and '"value"' not in text.split("EXCHANGE")[0][-400:]


# ---- round 6: one GPU ends in C3's RESULT; the N > 1 line explains its own scaling ---------------------------

def test_full_pass_of_all_eight_chunks_equals_the_eight_rank_step(tmp_path):
    """The timed region of the 1-GPU line cycles the chunks that fit in HBM; `c3_full` then accumulates ALL 8 chunks (problems 0 .. 8 Pc - 1,
    the sum o1.py:236-245 takes over every problem) into one buffer.  It must be, word for word (sha256 of the 8216 counters and the dump),
    the all-reduced step of 8 ranks x Pc problems over the same global problems -- here with fewer resident slots than chunks, so that
    several chunks are generated into freed slots -- and satisfy the closed forms of D2 / D5."""
    shape = ["--problems-per-step", "24", "--samples", str(1 << 13), "--no-cpu-baseline"]
    one = _bench([*shape, "--steps", "2", "--warmup", "1", "--dump", str(tmp_path / "one.npz")])
    f = one["c3_full"]
    assert f["problems"] == 192 and f["chunks"] == 8 and f["chunks_generated_after_the_timed_region"] == [3, 4, 5, 6, 7] and f["counters_words"] == 8 * 1027
    eight = _bench([*shape[:4], "--steps", "1", "--warmup", "0", "--dump", str(tmp_path / "eight.npz")], ranks=8)
    assert eight["c3_full"] is None and eight["counters_sha256_last_step"] == f["counters_sha256"]
    full = np.load(tmp_path / "one.npz")["full"]
    assert np.array_equal(full, np.load(tmp_path / "eight.npz")["counters"]) and full[:8 * 1025].sum() > 0
    assert [round(a, 6) for a in f["accuracy"]] == eight["accuracy_last_step"]
    # ... and one rank x 192 problems in ONE chunk
    whole = _bench(["--problems-per-step", "192", "--samples", str(1 << 13), "--no-cpu-baseline", "--steps", "1", "--warmup", "0", "--no-full-pass"])
    assert whole["c3_full"] is None and whole["counters_sha256_last_step"] == f["counters_sha256"]
    for dist, wins, tcs in ((2, 192, 192 << 13), (5, 0, 0)):
        d = _bench([*shape, "--steps", "1", "--warmup", "1", "--dist", str(dist)])["c3_full"]
        assert d["closed_form"].startswith(f"D{dist}:") and d["strict_wins"] == [wins] * 8 and d["truth_count_sum"] == [tcs] * 8
        assert d["accuracy"] == [1.0 if dist == 2 else 0.0] * 8


def test_multi_rank_line_reports_kernel_balance_and_the_exposed_all_reduce():
    """A first multi-GPU run below 6x must explain itself: per-rank kernel time min / max and the exposed all-reduce are in the torch line too."""
    d = _bench(["--problems-per-step", "64", "--samples", str(1 << 14), "--steps", "3", "--warmup", "1", "--resident", "2"], ranks=2)
    r = d["roofline"]
    assert 0 < r["kernel_avg_ms_per_rank_min"] <= r["kernel_avg_ms_per_rank_max"] == r["kernel_avg_ms"]
    assert r["exposed_allreduce_us"] is not None and r["exposed_allreduce_us"] > 0 and "AFTER the timed region" in r["exposed_allreduce_what"]
    assert d["allreduce_verified"] is True and d["warnings"] is None
    one = _bench(["--problems-per-step", "64", "--samples", str(1 << 14), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert one["roofline"]["exposed_allreduce_us"] is None and one["roofline"]["kernel_avg_ms_per_rank_min"] == one["roofline"]["kernel_avg_ms"]
    assert one["roofline"]["read_ceiling_warning"] is None and "frac_of_measured_read_ceiling" in one["roofline"]
