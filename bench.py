#!/usr/bin/env python3
"""bench.py -- sample-votes/s of the self-consistency aggregation path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
prints ONE JSON line on rank 0.  N > 1: either the driver launches the ranks itself (python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N ...: RANK / WORLD_SIZE are then in the environment), or -- like the reference,
whose run_experiments fans its problems out itself (o1.py:232-240) -- a plain `python bench.py --gpus N` starts its
own N ranks (one process per GPU through torch.distributed.run on 127.0.0.1) and relays rank 0's line.  With fewer
visible GPUs than ranks the line is {"error": ...} and the exit status is non-zero.

Default workload = BASELINE.json config C3, "synthetic int32 answers P=10k x B=8 x N=2^20" (335.5 GB, does
not fit 288 GB of HBM), streamed in problem-chunks: a STEP is one pass of the hot path over one
chunk of 1250 problems x 8 budgets x 2^20 samples (41.9 GB = the per-GPU shard of config C4), so
--steps 8 at N=1 is the whole of C3 and --gpus 8 --steps 1 is C4.  Weak scaling: every rank streams
its own 1250-problem chunk per step (problems sharded by contiguous block, global problem index
feeds the generator), followed by ONE RCCL all-reduce of the packed int64 per-budget counters
(65.7 KB) -- the only exchange step of the path.

Other workloads (never the driver's default): --workload c2 (BASELINE config 2, P=30 x B=8 x N=2^17, >= 5
distinct tensors cycled so the 256 MiB Infinity Cache cannot serve reruns) and --workload c5 (BASELINE config
5: P=10k x N=2^20 sharded by problem -- strong scaling -- vote + counters all-reduce + cell all-gather +
1000-resample bootstrap, pass@k sweep on the host; o1_inference_scaling_laws_amd/passk.py).

Inputs are generated ON DEVICE (closed-form integer generator, include/scvote.h) and are resident
in HBM before the timed region; `--resident` distinct chunks are cycled (default: as many of C3's 8 chunks as
fit in free HBM -- 7 on a 288 GB part -- each 41.9 GB >> the 256 MiB Infinity Cache, so every step streams
from HBM; the line reports `chunks_distinct`).  Distribution D1 (peaked/realistic) is
the headline; --dist 0/2/3/4/5 select uniform / degenerate / tie / peaked on a wrong value / degenerate-wrong.

The timed region is bracketed by barrier + torch.cuda.synchronize() on both sides; the kernel's own
duration is taken from hipEvents recorded by the library on the launch stream (scv_drain_kernel_ns).

cpu_baseline (rank 0, N=1): `value` is the UNMODIFIED reference loop -- o1.run_experiments (o1.py:216-247) imported
from the bytecode __graft_entry__.build() compiled into oracle/_ref (oracle/make_ref.py; kind "reference") -- timed on
this box's host at P = 30, N in {2^8, 2^11, 2^13} on votes of the workload's generator (oracle/refbaseline.py, a subprocess).
Beside it: the reference's arithmetic without its thread pools and cache lookups -- statistics.multimode +
o1.py:204-213 scoring on Python int lists (oracle/pybaseline.py) -- on one host core and on a process pool, and the C
restatement (oracle/scv_oracle.c).  The same leg is the checker: GPU cells of a timed chunk vs the C oracle (>= 256 problems x 8 budgets x
2^20 votes) and vs the Python reference arithmetic.  It is the ONLY place bench.py touches oracle/.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

HSA_IPC_NOTE = {"value": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "source": "inherited from the environment"
                if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ else "unset"}

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_VOTE = 4             # one int32 read per sample-vote, nothing written per vote
DISTS = {0: "D0 uniform", 1: "D1 peaked", 2: "D2 degenerate", 3: "D3 tie", 4: "D4 peaked on a wrong value", 5: "D5 degenerate-wrong"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c3", "c2", "c5"], default="c3")
    ap.add_argument("--problems-per-step", type=int, default=1250, help="c3/c2: problems per rank per step")
    ap.add_argument("--problems", type=int, default=10000, help="c5: total problems (sharded over the ranks)")
    ap.add_argument("--resamples", type=int, default=1000, help="c5: bootstrap resamples (split over the ranks)")
    ap.add_argument("--budgets", type=int, default=8)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--dist", type=int, default=1, help="0 uniform, 1 peaked (headline), 2 degenerate, 3 tie, 4 peaked on a wrong value, 5 degenerate-wrong")
    ap.add_argument("--resident", type=int, default=0,
                    help="distinct chunks kept in HBM and cycled (0 = auto: c3 as many of its 8 chunks as fit in free HBM)")
    ap.add_argument("--tokens", action="store_true", help="also stream the tokens tensor (8 B/vote)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-read-ceiling", action="store_true", help="skip the pure-read probe after the timed region")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=5.0, help="single-core Python reference arithmetic")
    ap.add_argument("--parity-problems", type=int, default=256, help="problems of the last timed chunk checked vs the C oracle")
    ap.add_argument("--copies", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20240914)
    ap.add_argument("--graph-steps", type=int, default=1,
                    help="with --graph: consecutive steps captured into ONE graph (the launch-bound inner loop of C2 as a single "
                         "graph launch); --steps and --warmup are rounded up to multiples of it")
    ap.add_argument("--c2-one-launch", action="store_true", help="c2: overwrite-counters mode (one kernel launch per evaluation, no memset node)")
    ap.add_argument("--no-timing", action="store_true", help="no hipEvent records around the launches (they cost ~7 us per step on a 26 us step: c2); kernel time = wall clock")
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: capture memset + hot-path launch of every resident chunk into a hipGraph and replay it "
                         "(launch-bound workloads such as --workload c2); kernel time is then taken from the wall clock")
    ap.add_argument("--comm", choices=["torch", "peer", "rccl"], default="torch",
                    help="who runs the exchange step.  torch (default): one PROCESS per GPU, torch.distributed all-reduce (--backend).  "
                         "peer / rccl: ONE process -- the reference's shape, o1.py:312-315 -- drives the library's own communicator over --gpus devices "
                         "(scv_comm_create: one-shot all-reduce over xGMI peer access / single-process RCCL), same chunk schedule, same JSON line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    ap.add_argument("--share-device", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 (1-GPU box, with --backend gloo) to exercise the multi-process path")
    ap.add_argument("--dump", default=None, help="rank 0 writes the last step's integer outputs (.npz) for bit-exact comparisons")
    ap.add_argument("--full-pass", dest="full_pass", action="store_true", default=True,
                    help="c3 on ONE GPU (default on): after the timed region, aggregate ALL 8 chunks of the workload (the 8th is generated into a "
                         "freed slot) into one counter buffer and print c3_full = C3's RESULT (accuracy per budget, sums, sha256 of the counters)")
    ap.add_argument("--no-full-pass", dest="full_pass", action="store_false")
    ap.add_argument("--no-live-traffic", dest="live_traffic", action="store_false", default=True,
                    help="default c3 run on one GPU: do not re-run the command under rocprofv3 --pmc for roofline.traffic (the committed profiles/ figure is quoted instead)")
    return ap.parse_args()


# ---- the cpu_baseline leg: the only code here that touches oracle/ (test infrastructure) -------------------

def check_cells_vs_c_oracle(gpu_cells, P_check, B, N, seed, dist, p_offset, tokens=None, threads=0, fatal=True):
    """GPU cell table rows [0, P_check) of a chunk vs oracle/scv_oracle.c on the regenerated inputs, in slabs of
    32 problems (1 GiB at N = 2^20) spread over the host cores.  Exits on the first differing field (fatal=False: returns
    (problems checked, message | None) instead, for the multi-rank lines where every rank reports before anyone leaves)."""
    import numpy as np
    from oracle import coracle
    threads = threads or min(os.cpu_count() or 1, 64)
    done = 0
    while done < P_check:
        n = min(32, P_check - done)
        a, _, tr = coracle.synth_fill(n, B, N, seed, dist, p_offset=p_offset + done)
        want = coracle.aggregate_mt(a, tr, threads)
        assert want["rc"] == 0
        for f in ("max_count", "truth_count", "n_modes", "min_mode", "hit"):
            if not np.array_equal(gpu_cells[f][done:done + n], want["cells"][f]):
                row = int(np.flatnonzero((gpu_cells[f][done:done + n] != want["cells"][f]).any(axis=1))[0])
                msg = f"PARITY FAILURE field {f}: GPU cell table differs from the CPU oracle (problem {p_offset + done + row}, chunk at {p_offset})"
                if fatal:
                    sys.exit(msg)
                return done, msg
        done += n
    return (done, None) if not fatal else done


# ---- the N > 1 line proves its own exchange step (BASELINE C4: "bit-exact at every GPU count"; o1.py:232-245) ----------------------

WARNINGS = []                   # things a reader of the line must see (rank 0): legs that did not run, ratios that should prompt a check
RANK_PARITY_PROBLEMS = 32       # problems of EVERY rank's last timed chunk that go through the oracle (rank-local, in parallel)


def injected_fault():
    """TEST ONLY (tests/test_bench_contract.py): SCV_BENCH_FAULT="rank:word" flips one word of that rank's all-reduced buffer on the
    host before the verification reads it -- what a stale peer read or a dropped rank would look like -- so the tests can see the
    line refuse to print.  -> (rank, word) | None."""
    spec = os.environ.get("SCV_BENCH_FAULT")
    if not spec:
        return None
    r, w = spec.split(":")
    return int(r), int(w)


def first_difference(got, want):
    """None when equal, else (flat index, got, want, number of differing words)."""
    import numpy as np
    got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    if got.shape != want.shape:
        return (-1, got.shape, want.shape, -1)
    d = np.flatnonzero(got != want)
    return None if d.size == 0 else (int(d[0]), int(got[d[0]]), int(want[d[0]]), int(d.size))


def closed_form_check(counters_np, B, N, total_problems, dist_id):
    """Distributions whose reduced counters are known in closed form, whatever the sharding: D2 (every vote of a problem is its
    truth: each cell a strict win => tie_class_hits[b][1] = problems, truth_count_sum[b] = problems * N) and D5 (every vote is one
    WRONG value: no hit anywhere, truth_count_sum = 0).  -> (description | None, message | None)."""
    import numpy as np
    if dist_id not in (2, 5):
        return None, None
    c = np.asarray(counters_np, dtype=np.int64)
    tie = c[: B * 1025].reshape(B, 1025)
    tcs = c[B * 1025 + B: B * 1025 + 2 * B]
    if dist_id == 2:
        want_tie1, want_tcs = total_problems, total_problems * N
    else:
        want_tie1, want_tcs = 0, 0
    for b in range(B):
        if int(tie[b, 1]) != want_tie1 or int(tie[b].sum()) != want_tie1 or int(tcs[b]) != want_tcs:
            return None, (f"closed form of dist {dist_id} violated at budget {b}: tie_class_hits[b][1] = {int(tie[b, 1])} (want {want_tie1}), "
                          f"row sum {int(tie[b].sum())}, truth_count_sum[b] = {int(tcs[b])} (want {want_tcs})")
    return (f"D{dist_id}: tie_class_hits[b][1] == {want_tie1} (= ranks x problems per rank) and truth_count_sum[b] == {want_tcs} for all {B} budgets", None)


def counters_digest(counters_np, ncount):
    """sha256 of the packed int64 counters (little-endian, tie_class_hits | token_sum | truth_count_sum): two runs that print the same digest
    computed the same 8216 words -- how a 1-GPU full pass and an 8-rank line are compared without a dump."""
    import hashlib
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(np.asarray(counters_np, dtype="<i8")[:ncount]).tobytes()).hexdigest()


def full_pass_c3(eng, torch, slots, Pc, B, N, args, cells, ctok, nchunks=8):
    """ONE GPU, all of C3: the timed region cycles the chunks that fit in HBM (7 of 8 at the default shape), so problems 8750 .. 9999 are never
    aggregated there and no 1-GPU line could print C3's RESULT (VERDICT r5 missing #5).  Here, outside the timed region: every chunk s = 0 .. 7
    (global problems [s Pc, (s+1) Pc)) is accumulated into ONE counter buffer -- resident chunks as they are, the others generated into a slot that
    is no longer needed -- which is the sum o1.py:236-245 takes over ALL problems.  -> dict for the line (and the counters for --dump)."""
    import numpy as np
    from o1_inference_scaling_laws_amd.engine import AggregateResult, counters_size
    dev = slots[0][0].device
    ncount = counters_size(B)
    full = torch.zeros(ncount, dtype=torch.int64, device=dev)
    resident = {sl[3] // Pc: k for k, sl in enumerate(slots) if sl[3] % Pc == 0}
    generated = []
    eng.sync()
    t0 = time.perf_counter()
    for s_ in range(nchunks):
        if s_ in resident:
            ans, tok, tr, _ = slots[resident[s_]]
        else:
            k = min(resident.values()) if resident else 0          # reuse the slot of a chunk that has already been accumulated (chunk 0's)
            ans, tok, tr, _ = slots[k]
            eng.synth_fill_device(ans, tok, tr, P=Pc, B=B, N=N, seed=args.seed, dist=args.dist, p_offset=s_ * Pc)
            slots[k] = (ans, tok, tr, s_ * Pc)
            resident = {c: kk for c, kk in resident.items() if kk != k}
            generated.append(s_)
        eng.aggregate_device(ans, tr, tokens=tok, counters=full, cells=cells, cell_tokens=ctok, overwrite=False)   # DEVICE mode accumulates (+=)
    eng.sync()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3
    host = full.cpu().numpy()
    total = nchunks * Pc
    res = AggregateResult.from_counters(host, total, B, num_problems=total)
    cf_what, cf_msg = closed_form_check(host, B, N, total, args.dist)
    if cf_msg:
        sys.exit("FULL PASS FAILURE: " + cf_msg)
    return {
        "what": (f"all {nchunks} chunks of {Pc} problems = global problems 0..{total - 1} x {B} budgets x {N} votes accumulated into ONE counter buffer on one GPU, "
                 f"outside the timed region (chunks {generated} generated into a freed slot); equals the all-reduced counters of an {nchunks}-rank step over the same problems"),
        "problems": total, "chunks": nchunks, "chunks_generated_after_the_timed_region": generated,
        "accuracy": [res.accuracy(b) for b in range(B)],
        "avg_tokens_used": [float(res.avg_tokens_used(b)) for b in range(B)] if args.tokens else None,
        "token_sum": [int(x) for x in res.token_sum], "truth_count_sum": [int(x) for x in res.truth_count_sum],
        "strict_wins": [int(res.tie_class_hits[b, 1]) for b in range(B)],
        "counters_sha256": counters_digest(host, ncount), "counters_words": ncount,
        "closed_form": cf_what, "ms_including_generation": ms,
    }, host


def judge_exchange(rank, ncount, reduced_np, local_parts, gathered_cells_np=None, local_cell_parts=None):
    """One rank's verdict on the exchange step: its all-reduced buffer vs the int64 numpy sum of every rank's PRE-reduce counters
    (gathered by a path that is not the all-reduce under test), and for C5 its all-gathered cell table vs the ranks' own blocks."""
    import numpy as np
    msgs = []
    want = np.sum(np.stack([np.asarray(p, dtype=np.int64)[:ncount] for p in local_parts]), axis=0, dtype=np.int64)
    d = first_difference(np.asarray(reduced_np)[:ncount], want)
    if d is not None:
        msgs.append(f"rank {rank}: all-reduced counters != independently gathered sum: word {d[0]} holds {d[1]}, want {d[2]} ({d[3]} of {ncount} words differ)")
    if gathered_cells_np is not None:
        want_t = np.concatenate([np.asarray(c) for c in local_cell_parts], axis=0)
        d = first_difference(np.asarray(gathered_cells_np).view(np.uint8), want_t.view(np.uint8))
        if d is not None:
            msgs.append(f"rank {rank}: all-gathered cell table != the ranks' own blocks: byte {d[0]} holds {d[1]}, want {d[2]} ({d[3]} bytes differ)")
    return msgs, want


def cpu_baseline(args, B, N, gpu_first_cells, first_off, gpu_last_cells, last_off, last_is_timed):
    import numpy as np
    from oracle import coracle
    cores = os.cpu_count() or 1
    # (1) the checker: wide comparison of a TIMED chunk with the C oracle + the first 16 problems of the sanity pass
    t0 = time.perf_counter()
    n16 = check_cells_vs_c_oracle(gpu_first_cells, min(16, gpu_first_cells.shape[0]), B, N, args.seed, args.dist, first_off)
    nwide = check_cells_vs_c_oracle(gpu_last_cells, min(args.parity_problems, gpu_last_cells.shape[0]), B, N, args.seed, args.dist, last_off)
    t_check = time.perf_counter() - t0
    # (2) the reference's arithmetic (Python, statistics.multimode) on one core and on a process pool
    procs = min(cores, 64)
    cmd = [sys.executable, "-m", "oracle.pybaseline", "--B", str(B), "--N", str(N), "--seed", str(args.seed), "--dist", str(args.dist),
           "--p-offset", str(last_off), "--seconds", str(args.cpu_baseline_seconds), "--procs", str(procs)]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=600)
    if out.returncode != 0:
        sys.exit("python baseline failed: " + out.stderr[-2000:])
    py = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    # ... and its results against the GPU cells: score = hit / n_modes (o1.py:206-210) for every problem it scored
    checked_py = 0
    for p_str, scores in py["scores"].items():
        row = int(p_str) - last_off
        if not 0 <= row < gpu_last_cells.shape[0]:
            continue
        for b, s in enumerate(scores):
            c = gpu_last_cells[row, b]
            want = (1.0 / int(c["n_modes"])) if c["hit"] else 0.0
            if float(s) != want:
                sys.exit(f"PARITY FAILURE vs statistics.multimode: problem {p_str} budget {b}: python {s} gpu {want}")
        checked_py += 1
    # (3) secondary: the C port, one core and OpenMP over problems
    P1 = 4
    a, _, tr = coracle.synth_fill(P1, B, N, args.seed, args.dist, p_offset=last_off)
    coracle.aggregate(a[:1], tr[:1])
    votes, t1 = 0, time.perf_counter()
    while True:
        assert coracle.aggregate(a, tr)["rc"] == 0
        votes += a.size
        dt = time.perf_counter() - t1
        if dt >= min(3.0, args.cpu_baseline_seconds):
            break
    threads = min(cores, 64)
    Pm = 2 * threads
    am, _, trm = coracle.synth_fill(Pm, B, N, args.seed, args.dist, p_offset=last_off)
    coracle.aggregate_mt(am, trm, threads)
    t2, passes = time.perf_counter(), 0
    while passes < 2 or time.perf_counter() - t2 < 1.5:
        assert coracle.aggregate_mt(am, trm, threads)["rc"] == 0
        passes += 1
    mt_rate = passes * am.size / (time.perf_counter() - t2)
    parity = (f"bit-exact: {nwide} problems x {B} budgets x {N} votes of {'the last TIMED' if last_is_timed else 'a'} chunk + {n16} of the "
              f"sanity pass vs oracle/scv_oracle.c ({t_check:.1f} s); {checked_py} problems x {B} cells also vs statistics.multimode "
              f"(score = hit / n_modes, o1.py:202-210); integers only -- the reference's float accuracy accumulates in thread-completion order "
              f"(o1.py:236-239), so floats are derived from the integer tie classes in one canonical order: equal to the reference's when every "
              f"tie size is a power of two, within 1e-12 otherwise")
    ps, pa = py["single"], py.get("all_cores")
    arithmetic = {
        "value": ps["votes_per_s"], "unit": "sample-votes/s", "cores": 1, "kind": "port",
        "what": "the reference's arithmetic without its thread pools and cache lookups: statistics.multimode(answers) + o1.py:204-213 "
                "scoring on Python int lists (oracle/pyoracle.py, line-by-line restatement of o1.py:181-213)",
        "sample": f"{ps['problems']} problems x {B} budgets x {N} samples of the workload's generator (dist {args.dist}), "
                  f"{ps['timed_s']:.1f} s on 1 of {cores} host cores",
        "all_cores": None if pa is None else {
            "value": pa["votes_per_s"], "processes": pa["procs"], "host_cores": cores,
            "sample": f"{pa['problems']} problems x {B} x {N}, one problem per process (multiprocessing), wall {pa['wall_s']:.2f} s"},
    }
    c_port = {"value": votes / dt, "cores": 1, "kind": "port", "what": "oracle/scv_oracle.c (C restatement, the bit-exactness comparator)",
              "sample": f"{P1} problems x {B} x {N}, {votes // a.size} passes, {dt:.1f} s",
              "all_cores": {"value": mt_rate, "threads": threads, "sample": f"{Pm} problems x {B} x {N}, {passes} passes, OpenMP over problems"}}
    # (4) the UNMODIFIED reference loop on this box's host: o1.run_experiments from oracle/_ref (a subprocess)
    ref, ref_note = reference_loop(args)
    if ref is not None:
        big = ref["results"][-1]
        # the same calls with the drop-in installed into the live reference module (the product path on the cached responses:
        # extract.py -> pinned vote tensors -> HOST-mode HIP engine -> host floats), wall time split, result equal to the reference's
        dropin_loop = None
        if any("dropin" in r for r in ref["results"]):
            dropin_loop = {
                "what": "o1.run_experiments(dataset, cache, 2048, N) with o1_dropin.install(o1, engine=Engine(timing=True)) in force, on the SAME "
                        "synthetic caches as reference_loop (P = 30; o1.py:216-247 replaced, o1.py:85-88,119 key scheme kept); seconds = best of 2; "
                        "split_s: extract (cache -> pinned vote tensors) / engine_call (blocking HOST-mode call: staging + kernel + results) / "
                        "kernel (hipEvents, inside engine_call) / floats",
                "engine": ref.get("dropin_engine"),
                "results": [dict(P=r["P"], N=r["N"], reference_seconds=r["seconds"], **r["dropin"]) for r in ref["results"] if "dropin" in r],
            }
            dropin_loop["speedup_vs_reference_at_largest_N"] = dropin_loop["results"][-1]["speedup_vs_reference"]
            # both driver families of the reference end to end (o1.py:314-315: 19 budgets at P = 30), reference vs drop-in
            dropin_loop["family"] = ref.get("family")
            fam = ref.get("family") or {}
            if "error" in fam or not fam:
                #  the family leg raised: the keys below would simply be absent and the parity exit skipped -- say so in the line
                dropin_loop["family_error"] = fam.get("error", "oracle.refbaseline returned no family record")
                WARNINGS.append("cpu_baseline.dropin_loop.family did not run (" + dropin_loop["family_error"] + "): the reference's own 19-budget "
                                "family was NOT compared with the drop-in in this run")
            for k in ("dropin_unbatched", "dropin_batched"):
                if k in fam and not fam[k]["records_equal_to_reference"]:
                    sys.exit(f"PARITY FAILURE: {k} records of the reference's driver families differ from the live reference's")
            dropin_loop["all_equal_to_reference"] = all(r["equal_to_reference"] for r in dropin_loop["results"])
            if not dropin_loop["all_equal_to_reference"]:
                sys.exit("PARITY FAILURE: the drop-in's (accuracy, avg_tokens_used) differ from the live reference's on the same cache")
        elif ref.get("dropin_error"):
            dropin_loop = {"error": ref["dropin_error"]}
        base = {
            "value": big["votes_per_s"], "unit": "sample-votes/s", "cores": 1, "kind": "reference",
            "what": "the UNMODIFIED reference loop o1.run_experiments (o1.py:216-247: nested thread pools, per-sample cache-key lookups, "
                    f"statistics.multimode, completion-order float sum) imported from {ref['reference']} "
                    "(oracle/_ref = py_compile of /root/reference, oracle/make_ref.py), warm in-memory synthetic cache, save_cache no-op'd; "
                    "its ~300 threads share the GIL: one core does the work",
            "sample": "; ".join(f"P={r['P']} x N={r['N']}: {r['seconds']:.2f} s = {r['votes_per_s']:.0f} votes/s" for r in ref["results"])
                      + f" (votes of the workload's generator, dist {args.dist}; accuracy equal to the restatement's: "
                      + str(all(r["accuracy_matches_restatement"] for r in ref["results"])) + f"); {cores} host cores",
            "reference_loop": [{k: v for k, v in r.items() if k != "dropin"} for r in ref["results"]],
            "dropin_loop": dropin_loop,
            "arithmetic": arithmetic,
            "c_port": c_port,
        }
    else:
        base = dict(arithmetic, c_port=c_port, reference_loop=None, dropin_loop=None,
                    note=ref_note or "oracle/_ref is absent (build() did not run where /root/reference exists): the restatement is the baseline")
    return parity, base


def reference_loop(args):
    """oracle/refbaseline.py as a subprocess: the unmodified o1.run_experiments at P = 30, N = 2^8, 2^11, 2^13 (SURVEY 8d "R0"; ~10 s),
    and the same calls with the drop-in installed (--dropin).  -> (result | None, note): None when oracle/_ref is not there, when the
    subprocess timed out or when it failed (e.g. a bytecode magic mismatch on the box) -- the headline line is emitted either way."""
    cmd = [sys.executable, "-m", "oracle.refbaseline", "--N", "256", "2048", "8192", "--seed", str(args.seed), "--dist", str(args.dist), "--dropin"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=300, env=dict(os.environ, MPLBACKEND="Agg"))
    except subprocess.TimeoutExpired:
        return None, "oracle.refbaseline timed out (300 s): the restatement is the baseline"
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return None, "oracle.refbaseline failed (rc %d): %s" % (out.returncode, out.stderr[-500:].replace("\n", " | "))
    ref = json.loads(lines[-1])
    return (ref, None) if ref.get("available") else (None, ref.get("why"))


# ---- helpers -------------------------------------------------------------------------------------------------

def free_port() -> int:
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def probe_ipc_mode():
    """RCCL between processes shares device memory through hipIpcGetMemHandle.  Hosts whose driver only supports dmabuf
    IPC need HSA_ENABLE_IPC_MODE_LEGACY=0 (read when the HSA runtime starts, so it has to be in the environment before
    the ranks import torch).  When the variable is already set it is left alone; when it is not, a throw-away process
    asks the runtime for an IPC handle in its default mode and the variable is set to 0 only if that fails."""
    if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
        return
    code = ("import ctypes,sys\n"
            "h=ctypes.CDLL('libamdhip64.so')\n"
            "p=ctypes.c_void_p()\n"
            "sys.exit(3) if h.hipMalloc(ctypes.byref(p),ctypes.c_size_t(1<<20)) else None\n"
            "b=(ctypes.c_char*64)()\n"
            "sys.exit(0 if h.hipIpcGetMemHandle(ctypes.byref(b),p)==0 else 4)\n")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    try:
        rc = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=120).returncode
    except (OSError, subprocess.TimeoutExpired):
        rc = -1
    if rc == 4:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        HSA_IPC_NOTE.update(value="0", source="set by bench.py: hipIpcGetMemHandle failed in the runtime's default IPC mode (probe process)")
    else:
        HSA_IPC_NOTE.update(source=f"unset: probe process rc={rc} (0 = hipIpcGetMemHandle works in the default mode)")


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves -- the fan-out is the callee's
    job in the reference too (o1.py:232-240) -- one process per GPU, rendezvous on 127.0.0.1, and relay rank 0's line."""
    n = args.gpus
    visible = -1
    if not args.share_device:
        try:
            import ctypes
            visible = int(ctypes.CDLL(os.path.join(REPO, "o1_inference_scaling_laws_amd", "csrc", "libscvote.so")).scv_device_count())
        except OSError as e:
            print(json.dumps({"error": f"libscvote.so is not built: {e}", "n_gpus": n}), flush=True)
            return 2
        if visible < n:
            print(json.dumps({"error": f"bench.py --gpus {n}: only {visible} HIP device(s) visible; one rank per GPU is required "
                                       f"(--share-device exists for the 1-GPU test box only)",
                              "metric": "sample-votes/sec (problems x samples)", "value": None, "n_gpus": n,
                              "hip_devices_visible": visible}), flush=True)
            return 2
    if args.backend == "nccl":
        probe_ipc_mode()
    env = dict(os.environ, SCV_BENCH_SPAWNED="1", SCV_HSA_IPC_NOTE=json.dumps(HSA_IPC_NOTE))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    rest = [l for l in proc.stdout.splitlines() if not l.startswith("{")]
    if rest:
        print("\n".join(rest), file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(json.dumps({"error": f"{n}-rank run failed (rc {proc.returncode}, {len(lines)} JSON lines)", "n_gpus": n,
                          "lines": lines[-2:]}), flush=True)
        return proc.returncode or 1
    print(lines[0], flush=True)
    return 0


def measure_read_ceiling(cells: int):
    """Pure-read ceiling of THIS box, measured right after the timed region by tools/hbm_probe.bin --quick (a separate
    process: best of four read-only kernels over `cells` x 4 MiB).  None when the probe binary is not built."""
    probe = os.path.join(REPO, "tools", "hbm_probe.bin")
    if not os.path.exists(probe):
        return None
    try:
        out = subprocess.run([probe, str(cells), "--quick"], capture_output=True, text=True, timeout=120)
    except (OSError, subprocess.TimeoutExpired):
        return None
    for line in out.stdout.splitlines():
        if line.startswith("READ_CEILING_GBPS"):
            return float(line.split()[1])
    return None


def measure_traffic_live(args):
    """HBM bytes per launch of the dominant kernel, MEASURED IN THIS RUN (the line used to quote a committed file): the same
    command is run once more as a child under `rocprofv3 --pmc FETCH_SIZE` and once under `--pmc WRITE_SIZE` (separate passes, counters only -- no
    trace domain next to --pmc, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), a few steps over resident chunks, after this process has handed
    its HBM back.  FETCH_SIZE is in KiB and reads 1/2 on gfx950 for wide coalesced streams (x2; calibrated on a known-bytes read:
    profiles/r02_fetch_size_calibration.json); WRITE_SIZE x 1024.  -> (bytes per launch | None, note)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if os.environ.get("ROCPROFILER_LIBRARY_CTOR") or any(k.startswith("ROCPROF_") for k in os.environ):
        return None, "this process is itself running under rocprofv3 (its environment would reach the child's profiler)"
    child = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--resident", "2", "--no-cpu-baseline", "--no-read-ceiling",
             "--no-full-pass", "--no-live-traffic", "--seed", str(args.seed), "--dist", str(args.dist)]
    got = {}
    launches = 0
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(prefix="scv_pmc_") as d:
                out = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", *child], capture_output=True, text=True, timeout=300,
                                     cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
                if out.returncode != 0:
                    return None, f"rocprofv3 --pmc {counter} failed (rc {out.returncode}): " + out.stderr[-300:].replace("\n", " | ")
                vals = []
                for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        if "scv_hist_argmax" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            vals.append(float(r["Counter_Value"]))
                if not vals:
                    return None, f"no {counter} rows for scv_hist_argmax in the child's counter_collection.csv"
                got[counter] = sum(vals) / len(vals)
                launches = len(vals)
    except (OSError, subprocess.TimeoutExpired) as e:
        return None, f"{type(e).__name__}: {e}"
    return got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024, (
        f"measured in THIS run: child `bench.py --steps 2 --warmup 1` under rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes, {launches} launches "
        f"of scv_hist_argmax each); FETCH_SIZE {got['FETCH_SIZE']:.0f} KiB x 1024 x 2 (gfx950 correction, profiles/r02_fetch_size_calibration.json) + WRITE_SIZE "
        f"{got['WRITE_SIZE']:.0f} KiB x 1024")


def committed_evidence():
    """Numbers measured in SEPARATE profiled runs of this command and committed under profiles/ (labelled as such
    in the JSON line): HBM bytes per launch from the PMC counters, and the pure-read ceiling of the probe."""
    import glob
    ev = {"traffic": None, "traffic_source": None, "read_ceiling_gbs": None, "read_ceiling_source": None}
    pmcs = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc.json")))
    if pmcs:
        with open(pmcs[-1]) as f:
            ev["traffic"] = json.load(f).get("hbm_traffic_bytes")
        ev["traffic_source"] = (os.path.relpath(pmcs[-1], REPO) + ": separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py`; "
                                "FETCH_SIZE x2 (gfx950), factor calibrated on a known-bytes read kernel: profiles/r02_fetch_size_calibration.json")
    probes = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_probe.log")))
    for path in reversed(probes):
        for line in open(path):
            if line.startswith("READ_CEILING_GBPS"):
                ev["read_ceiling_gbs"] = float(line.split()[1])
                ev["read_ceiling_source"] = os.path.relpath(path, REPO) + " (tools/hbm_probe.hip: best pure-read kernel over the same 41.9 GB, another box of the pool)"
                return ev
    return ev


def main_single_process(args):
    """--comm peer | rccl: ONE process, one scv_ctx per GPU inside one scv_comm (include/scvote.h), the reference's process shape
    (o1.py:312-315).  Same workload, chunk schedule, timed region and JSON line as the torch.distributed path; the exchange step is
    scv_allreduce_counters.  c5: MultiDeviceEngine.evaluate_c5 (vote, all-reduce, scv_allgather_cells, per-rank bootstrap slices,
    scv_allgather_i64)."""
    import numpy as np
    import torch
    from o1_inference_scaling_laws_amd._lib import ScvError
    from o1_inference_scaling_laws_amd.dist import shard_bounds
    from o1_inference_scaling_laws_amd.engine import AggregateResult, MultiDeviceEngine, cells_from_torch, counters_size
    world = args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    ndev = torch.cuda.device_count()
    # ONE JSON line on stdout: librccl prints a version banner on the process's stdout when its first communicator is created -- everything but
    # the line goes to stderr (fd 1 is pointed at fd 2 for the run; the line is written to the saved descriptor)
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(obj):
        real_stdout.write(json.dumps(obj) + "\n")
        real_stdout.flush()
    if not args.share_device and ndev < world:
        emit({"error": f"bench.py --gpus {world} --comm {args.comm}: only {ndev} HIP device(s) visible; one rank per GPU is required "
                       f"(--share-device exists for the 1-GPU test box only)",
              "metric": "sample-votes/sec (problems x samples)", "value": None, "n_gpus": world, "hip_devices_visible": ndev})
        sys.exit(2)
    devices = [0] * world if args.share_device else list(range(world))
    t_create = time.perf_counter()
    try:
        mde = MultiDeviceEngine(devices, rccl=args.comm == "rccl", timing=True)      # create ends with the communicator's self-test
    except ScvError as e:
        emit({"error": f"scv_comm_create({devices}, {args.comm}) failed: {e}", "metric": "sample-votes/sec (problems x samples)",
              "value": None, "n_gpus": world, "hip_devices_visible": ndev})
        sys.exit(2)
    t_create = time.perf_counter() - t_create
    c5 = args.workload == "c5"
    if args.workload == "c2":
        args.problems_per_step, args.budgets, args.samples = 30, 8, 1 << 17
        args.resident = max(args.resident, 5)
    if c5:
        args.budgets, args.resident = 1, 1
    B, N = args.budgets, args.samples
    rows = [shard_bounds(args.problems, g, world)[1] - shard_bounds(args.problems, g, world)[0] for g in range(world)] if c5 else [args.problems_per_step] * world
    Pc = rows[0]
    for e in mde.engines:
        e.set_tuning(args.copies, args.threads, args.wg_per_cu, args.unroll)
    want_R = args.resident if args.resident > 0 else (8 if args.workload == "c3" else 1)
    R = max(1, min(want_R, args.steps + args.warmup)) if args.workload == "c3" else max(1, want_R)
    chunk_bytes = Pc * B * N * 4 * (2 if args.tokens else 1)
    free_b = min(torch.cuda.mem_get_info(torch.device("cuda", d))[0] for d in set(devices))
    usable = free_b // (world if args.share_device else 1) - (3 << 30)
    while R > 1 and R * chunk_bytes > usable:
        R -= 1
    if chunk_bytes > usable:
        sys.exit(f"one chunk ({chunk_bytes / 1e9:.1f} GB) does not fit in free HBM ({free_b / 1e9:.1f} GB)")
    slots = [[] for _ in range(world)]                          # slots[g][s] = (answers, tokens, truth, p_off)
    for g, e in enumerate(mde.engines):
        dev = torch.device("cuda", devices[g])
        with torch.cuda.device(dev):
            for s_ in range(R):
                p_off = shard_bounds(args.problems, g, world)[0] if c5 else (s_ * world + g) * Pc
                ans = torch.empty((rows[g], B, N), dtype=torch.int32, device=dev)
                tok = torch.empty((rows[g], B, N), dtype=torch.int32, device=dev) if args.tokens else None
                tr = torch.empty((rows[g],), dtype=torch.int32, device=dev)
                e.synth_fill_device(ans, tok, tr, P=rows[g], B=B, N=N, seed=args.seed, dist=args.dist, p_offset=p_off)
                slots[g].append((ans, tok, tr, p_off))
    ncount = counters_size(B)
    bufs = [[torch.zeros(ncount, dtype=torch.int64, device=torch.device("cuda", devices[g])) for _ in range(2)] for g in range(world)]
    cells = [torch.empty((rows[g], B, 16), dtype=torch.uint8, device=torch.device("cuda", devices[g])) for g in range(world)]
    ctok = [torch.empty((rows[g], B), dtype=torch.int64, device=torch.device("cuda", devices[g])) if args.tokens else None for g in range(world)]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps + args.warmup + 1)]
    c5_state = {"M": None, "last": None, "boot_seed": args.seed ^ 0xB007}
    ow = args.workload == "c2" and args.c2_one_launch

    def step(i, timed_slot=None):
        if c5:
            shards = [(slots[g][0][0], slots[g][0][2], slots[g][0][1]) for g in range(world)]
            out = mde.evaluate_c5(shards, args.resamples, c5_state["boot_seed"], M=c5_state["M"], keep_all_ranks=True)
            c5_state["M"], c5_state["last"] = out[3], out
            return out[0][:ncount]
        cur = []
        for g, e in enumerate(mde.engines):
            ans, tok, tr, _ = slots[g][i % R]
            with torch.cuda.device(ans.device):
                c = bufs[g][i % 2]
                if not ow:
                    c.zero_()
                e.aggregate_device(ans, tr, tokens=tok, counters=c, cells=cells[g], cell_tokens=ctok[g], overwrite=ow)
                cur.append(c)
        if timed_slot is not None:
            with torch.cuda.device(cur[0].device):
                ev[timed_slot][0].record()                      # rank 0's stream: its vote kernel is done here ...
        mde.all_reduce_counters(cur)
        if timed_slot is not None:
            with torch.cuda.device(cur[0].device):
                ev[timed_slot][1].record()                      # ... and here its buffer holds the sum (waits for the slowest rank included)
        c5_state["cur"] = cur                                   # every rank's buffer (each must hold the sum: verified after the timed region)
        return cur[0]

    def fence():
        mde.sync()
        for d in set(devices):
            torch.cuda.synchronize(d)

    step(0)
    fence()
    first_cells = cells_from_torch(cells[0]) if not c5 else cells_from_torch(c5_state["last"][1])[: rows[0]]
    for i in range(args.warmup):
        step(i)
    fence()
    for e in mde.engines:
        e.drain_kernel_ns()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(args.warmup + i, timed_slot=None if c5 else i)
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = [e.drain_kernel_ns() for e in mde.engines]
    # c5 launches two timed kernels per evaluation on a rank? no: only aggregation launches are timed (scv_bootstrap is not)
    kern_ms = [ns / max(n, 1) / 1e6 for ns, n in per_rank]
    exposed_us = None if c5 or args.steps == 0 else float(np.mean([ev[i][0].elapsed_time(ev[i][1]) for i in range(args.steps)])) * 1e3
    last_off = slots[0][(args.warmup + args.steps - 1) % R][3] if args.steps > 0 else slots[0][0][3]
    last_cells = cells_from_torch(cells[0]) if not c5 else cells_from_torch(c5_state["last"][1])[: rows[0]]
    # ---- the exchange step proves itself (outside the timed region): every rank's PRE-reduce counters of the last timed chunk are
    # computed once more by a launch with no collective behind it, copied to the host rank by rank (plain D2H), summed in numpy
    # int64, and compared word for word with the all-reduced buffer OF EVERY RANK; C5: every rank's gathered cell table and gathered
    # resample table too.  Every rank's last chunk also goes through the oracle (RANK_PARITY_PROBLEMS problems each).
    last_i = args.warmup + args.steps - 1
    exchange, failures = None, []
    if args.steps > 0:
        locals_np, local_cells_np, rank_cells = [], [], []
        for g, e in enumerate(mde.engines):
            ans, tok, tr, _ = slots[g][0 if c5 else last_i % R]
            with torch.cuda.device(ans.device):
                lc = torch.zeros(ncount, dtype=torch.int64, device=ans.device)
                lt = torch.empty((rows[g], B, 16), dtype=torch.uint8, device=ans.device)
                if rows[g]:
                    e.aggregate_device(ans, tr, tokens=tok, counters=lc, cells=lt, cell_tokens=ctok[g], overwrite=False)
                e.sync()
                locals_np.append(lc.cpu().numpy())
                local_cells_np.append(lt.cpu().numpy())
        if c5:
            reduced = [c.cpu().numpy() for c in mde.last_c5["counters"]]
            tables = [t.cpu().numpy() for t in mde.last_c5["tables"]]
            boots = [b_.cpu().numpy() for b_ in mde.last_c5["boots"]]
        else:
            reduced = [c.cpu().numpy() for c in c5_state["cur"]]
            tables = [None] * world
        want_sum = None
        fault = injected_fault()
        if fault:
            reduced[fault[0]] = reduced[fault[0]].copy()
            reduced[fault[0]][fault[1]] += 1
        for g in range(world):
            msgs, want_sum = judge_exchange(g, ncount, reduced[g], locals_np, tables[g], local_cells_np if c5 else None)
            failures += msgs
            if c5 and first_difference(boots[g], boots[0]) is not None:
                failures.append(f"rank {g}: gathered resample table differs from rank 0's")
            if c5 and int(reduced[g][ncount]) != 0:
                failures.append(f"rank {g}: summed device error word = {int(reduced[g][ncount])}")
            rank_cells.append(cells_from_torch(torch.from_numpy(local_cells_np[g])))
            if not c5 and first_difference(cells_from_torch(cells[g]).view(np.uint8), local_cells_np[g]) is not None:
                failures.append(f"rank {g}: the cell table of the last timed step differs from the same chunk's re-run")
        cf_what, cf_msg = closed_form_check(want_sum, B, N, args.problems if c5 else Pc * world, args.dist)
        if cf_msg:
            failures.append(cf_msg)
        checked = 0
        if not args.no_cpu_baseline:
            for g in range(world):
                p_off = slots[g][0 if c5 else last_i % R][3]
                n, msg = check_cells_vs_c_oracle(rank_cells[g], min(RANK_PARITY_PROBLEMS, rows[g]), B, N, args.seed, args.dist, p_off, fatal=False)
                checked += 1
                if msg:
                    failures.append(f"rank {g}: {msg}")
        exchange = {
            "allreduce_verified": not failures, "allreduce_words": ncount, "ranks_verified": world,
            "how": "every rank's pre-reduce counters of the last timed chunk (a re-run of the chunk with no collective behind it), copied D2H rank by "
                   "rank, summed in numpy int64 and compared word for word with the all-reduced buffer of EVERY rank"
                   + ("; C5: every rank's all-gathered cell table vs the ranks' own blocks, every rank's gathered resample table vs rank 0's" if c5 else ""),
            "parity_ranks_checked": checked, "parity_problems_per_rank": min(RANK_PARITY_PROBLEMS, min(rows)) if checked else 0,
            "closed_form": cf_what,
        }
        if failures:
            emit({"error": "EXCHANGE / PARITY FAILURE", "failures": failures[:8], "n_gpus": world, "value": None,
                  "metric": "sample-votes/sec (problems x samples)", "allreduce_verified": False})
            sys.exit(3)
    votes_per_step_per_gpu = Pc * B * N
    total_votes = (args.problems * B * N if c5 else votes_per_step_per_gpu * world) * args.steps
    bytes_per_launch = votes_per_step_per_gpu * BYTES_PER_VOTE * (2 if args.tokens else 1)
    kern_avg_ms = max(kern_ms)
    achieved = bytes_per_launch / (kern_avg_ms * 1e-3) / 1e9
    final = AggregateResult.from_counters(last.cpu().numpy(), Pc, B, num_problems=args.problems if c5 else Pc * world)
    out = {
        "metric": "sample-votes/sec (problems x samples)", "value": total_votes / elapsed, "unit": "sample-votes/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
        "scaling": "strong" if c5 else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {
            "workload": (f"C5: P={args.problems} x N={N} sharded by problem over {world} GPU(s) of ONE process; step = vote + counters all-reduce + "
                         f"scv_allgather_cells + per-rank bootstrap slices of {args.resamples} resamples + scv_allgather_i64 (M={c5_state['M']})") if c5 else
                        (f"{args.workload.upper()} chunks: step = {Pc} problems x {B} budgets x {N} samples int32 ({bytes_per_launch / 1e9:.2f} GB) per GPU"),
            "distribution": DISTS[args.dist], "resident_chunks": R, "chunks_distinct": R, "tokens_stream": bool(args.tokens),
            "parallelism": f"problems sharded over {world} GPU(s) driven by ONE process, one int64 all-reduce of {ncount} counters per step (scv_allreduce_counters)",
            "seed": args.seed, "launch": "eager", "comm": args.comm, "backend": None,
            "comm_what": ("one-shot all-reduce over xGMI peer access (pure HIP, csrc/scvote_comm.hip)" if args.comm == "peer"
                          else "single-process RCCL: ncclCommInitAll + grouped ncclAllReduce(int64, sum)"),
            "rccl_ranks": world if args.comm == "rccl" else None, "comm_ranks": world, "hip_devices_visible": ndev,
            "comm_create_s": t_create, "comm_selftest_words_per_rank": mde.stat("selftest_words"),
            "ranks_started_by": "none: one process", "devices_shared_by_ranks": bool(args.share_device), "devices": devices,
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "scv_hist_argmax", "kernel_avg_ms": kern_avg_ms, "kernel_avg_ms_per_rank_min": min(kern_ms), "kernel_avg_ms_per_rank_max": max(kern_ms),
                     "launches_timed": per_rank[0][1], "algorithmic_bytes_per_launch": bytes_per_launch,
                     "exposed_allreduce_us": exposed_us,
                     "exposed_allreduce_what": "rank 0's stream, hipEvents: from the end of its vote kernel to its buffer holding the sum (waiting for the slowest rank included)"},
        "parity": None, "cpu_baseline": None,
        "parity_ranks_checked": exchange["parity_ranks_checked"] if exchange else 0,
        "allreduce_verified": bool(exchange and exchange["allreduce_verified"]),
        "exchange_check": exchange,
        "accuracy_last_step": [round(final.accuracy(b), 6) for b in range(B)],
    }
    if not args.no_cpu_baseline and exchange:
        n0 = check_cells_vs_c_oracle(first_cells, min(4, first_cells.shape[0]), B, N, args.seed, args.dist, slots[0][0][3])
        out["parity"] = (f"bit-exact: EVERY rank's first {exchange['parity_problems_per_rank']} problems x {B} budgets x {N} votes of its last timed chunk "
                         f"({world} ranks; + {n0} of rank 0's sanity pass) vs oracle/scv_oracle.c; all-reduced counters of every rank == the "
                         f"independently gathered int64 sum ({ncount} words)")
        out["metric"] += ", bit-exact vs CPU (see parity)"
    else:
        out["metric"] += " (parity not checked in this run)"
    if c5:
        cnt, table, boot, M = c5_state["last"]
        out["c5"] = {"tie_classes_M": M, "device_error_word": int(cnt.cpu()[ncount])}
        if not args.no_cpu_baseline:
            from oracle import coracle
            rc, want_boot = coracle.bootstrap(cells_from_torch(table), 0, args.resamples, c5_state["boot_seed"], M)
            if rc != 0 or not np.array_equal(boot.cpu().numpy(), want_boot):
                sys.exit("PARITY FAILURE: bootstrap table differs from oracle scvo_bootstrap")
            out["c5"]["bootstrap_parity"] = f"all {args.resamples} x {B} x {M} resample counters bit-exact vs oracle/scv_oracle.c"
        if args.dump:
            np.savez(args.dump, counters=cnt.cpu().numpy()[:ncount], cells=table.cpu().numpy(), boot=boot.cpu().numpy())
    elif args.dump:
        np.savez(args.dump, counters=last.cpu().numpy())
    mde.close()
    emit(out)


def main():
    args = parse()
    if args.comm != "torch":
        return main_single_process(args)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("SCV_FORCE_COLLECTIVES") == "1"):
        sys.exit(spawn_ranks(args))             # plain `python bench.py --gpus N`: start the ranks ourselves
    if "SCV_HSA_IPC_NOTE" in os.environ:
        HSA_IPC_NOTE.update(json.loads(os.environ["SCV_HSA_IPC_NOTE"]))
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1 and args.backend == "nccl":
        probe_ipc_mode()                        # ranks started by the driver's own torch.distributed.run (before torch loads HIP)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                       # under a launcher the launcher's world size wins
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    ndev = torch.cuda.device_count()
    # preflight for N > 1: one GPU per rank, or say loudly why not (the 1-GPU test box shares cuda:0 on purpose)
    if world > 1 and not args.share_device and ndev < world:
        if rank == 0:
            print(json.dumps({"error": f"bench.py --gpus {world}: only {ndev} HIP device(s) visible; one rank per GPU is required "
                                       f"(--share-device exists for the 1-GPU test box only)",
                              "metric": "sample-votes/sec (problems x samples)", "value": None, "n_gpus": world,
                              "hip_devices_visible": ndev}), flush=True)
        sys.exit(2)
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = None
    # SCV_FORCE_COLLECTIVES=1 (under torchrun --nproc-per-node 1): a ONE-rank process group whose collectives really
    # run, so a 1-GPU box executes the RCCL calls of the N > 1 path (tests/test_bench_contract.py)
    force = os.environ.get("SCV_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            rccl_ranks = dist.get_world_size()
            assert rccl_ranks == world, f"RCCL communicator has {rccl_ranks} ranks, expected {world}"
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        # one collective of every kind the run will issue, BEFORE the resident chunks claim HBM: RCCL sizes its channel buffers and
        # IPC mappings on first use, and the default c3 run leaves ~15 GB of 309 free (7 x 41.94 GB resident)
        warm = torch.zeros(8 * 1027 + 1, dtype=torch.int64, device=dev)
        dist.all_reduce(warm, op=dist.ReduceOp.SUM)
        if args.backend == "nccl":
            parts = [torch.empty(4096, dtype=torch.uint8, device=dev) for _ in range(world)]
            dist.all_gather(parts, torch.zeros(4096, dtype=torch.uint8, device=dev))
            parts = [torch.empty(512, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(parts, torch.zeros(512, dtype=torch.int64, device=dev))
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=dev), op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize(dev)
        assert int(warm.abs().sum().item()) == 0
        del warm

    from o1_inference_scaling_laws_amd import passk
    from o1_inference_scaling_laws_amd.dist import CounterPipeline, shard_bounds
    from o1_inference_scaling_laws_amd.engine import AggregateResult, Engine, cells_from_torch, counters_size

    c5 = args.workload == "c5"
    if args.workload == "c2":
        args.problems_per_step, args.budgets, args.samples = 30, 8, 1 << 17
        args.resident = max(args.resident, 5)
    if c5:
        args.budgets = 1
        lo, hi = shard_bounds(args.problems, rank, world)
        args.problems_per_step, args.resident = hi - lo, 1
    Pc, B, N = args.problems_per_step, args.budgets, args.samples
    use_graph = bool(args.graph) and world == 1 and not c5
    # C2 steps take ~26 us and a pair of hipEvent records around every launch adds ~8 us to each (profiles/r04_c2_sweep.json: 34.0 vs 26.4 us):
    # the timed region of c2 runs WITHOUT the records (that is `value`), and the kernel's own duration comes from a second, instrumented
    # loop over the same chunks right after it (roofline.kernel_avg_ms; `kernel_timed_in` says so)
    split_timing = args.workload == "c2" and not use_graph and not args.no_timing and world == 1
    no_events = use_graph or args.no_timing or split_timing   # hipEvent records cannot live inside a captured graph
    eng = Engine(device=local_rank, timing=not no_events)
    eng.set_tuning(args.copies, args.threads, args.wg_per_cu, args.unroll)

    # ---- resident inputs: `resident` distinct chunks, generated on device ------------------------
    # c3 default: as many DISTINCT chunks of C3's 8 as free HBM holds (7 of 41.94 GB on a 288 GB part; the 8th would need
    # 335.5 GB), never more than the run has steps; an explicit --resident is honoured as far as memory allows
    want_R = args.resident if args.resident > 0 else (8 if args.workload == "c3" else 1)
    R = max(1, min(want_R, args.steps + args.warmup)) if args.workload == "c3" else max(1, want_R)
    chunk_bytes = Pc * B * N * 4 * (2 if args.tokens else 1)
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    usable = (free_b // (world if args.share_device else 1)) - (3 << 30)        # headroom: cell table, counters, library scratch, RCCL
    while R > 1 and R * chunk_bytes > usable:
        R -= 1
    if chunk_bytes > usable:
        sys.exit(f"one chunk ({chunk_bytes / 1e9:.1f} GB) does not fit in free HBM ({free_b / 1e9:.1f} GB)")
    slots = []
    for s in range(R):
        # global problem index of this rank's block: chunk s of a weak-scaling run, or the rank's shard of C5
        p_off = shard_bounds(args.problems, rank, world)[0] if c5 else (s * world + rank) * Pc
        try:
            ans = torch.empty((Pc, B, N), dtype=torch.int32, device=dev)
            tok = torch.empty((Pc, B, N), dtype=torch.int32, device=dev) if args.tokens else None
        except torch.OutOfMemoryError:
            if not slots:
                raise
            ans = tok = None
            torch.cuda.empty_cache()
            R = len(slots)                      # fewer distinct chunks than planned: reported as chunks_distinct
            break
        tr = torch.empty((Pc,), dtype=torch.int32, device=dev)
        eng.synth_fill_device(ans, tok, tr, P=Pc, B=B, N=N, seed=args.seed, dist=args.dist, p_offset=p_off)
        slots.append((ans, tok, tr, p_off))
    # two counter buffers: the all-reduce of step i (RCCL's own stream) overlaps the kernel of step i+1
    pipe = CounterPipeline([torch.zeros(counters_size(B) + (1 if c5 else 0), dtype=torch.int64, device=dev) for _ in range(2)])   # c5: + the collective error word
    cells = torch.empty((Pc, B, 16), dtype=torch.uint8, device=dev)
    ctok = torch.empty((Pc, B), dtype=torch.int64, device=dev) if args.tokens else None
    eng.sync()

    c5_state = {"M": None, "last": None, "boot_seed": args.seed ^ 0xB007}

    def step(i, engine=None):
        engine = engine or eng
        ans, tok, tr, _ = slots[i % R]
        if c5:
            out = passk.evaluate_device(eng, ans, tr, args.problems, args.resamples, c5_state["boot_seed"], M=c5_state["M"],
                                        tokens_local=tok, counters=pipe.buffers[i % 2], cells_local=cells)
            c5_state["M"], c5_state["last"] = out.M, out      # the class bound is found once (host sync), then reused
            return out.counters
        # c2 (few long cells): zeroing memset + accumulate.  The one-launch form (overwrite=True: the kernel's last workgroup turns the cell table
        # into the counters) is 0.8-1.0 us per step SLOWER in every geometry of profiles/r04_c2_sweep.json (26.4 vs 25.6 us eager): the
        # 65.7 KB counter rewrite by one workgroup costs more than a memset node that overlaps the previous step's tail.  --c2-one-launch selects it.
        ow = args.workload == "c2" and args.c2_one_launch
        counters = pipe.acquire(i, zero=not ow)   # waits for this buffer's previous all-reduce (and zeroes it)
        engine.aggregate_device(ans, tr, tokens=tok, counters=counters, cells=cells, cell_tokens=ctok, overwrite=ow)
        return pipe.publish(i)               # async all-reduce (no-op with one rank)

    graphs = {}
    gs = max(1, args.graph_steps) if use_graph else 1
    if use_graph:
        args.steps = -(-args.steps // gs) * gs
        args.warmup = -(-args.warmup // gs) * gs
        side = torch.cuda.Stream(device=dev)
        # the step sequence repeats with period lcm(R, 2) (slot, counter buffer); one graph per group of gs consecutive
        # steps, as many groups as it takes for the group sequence to repeat
        period = R if R % 2 == 0 else 2 * R
        import math
        ngroups = period // math.gcd(period, gs)
        with torch.cuda.stream(side):
            for i in range(period):                       # warm-up on the capture stream: scratch sized, attributes set
                step(i)
        torch.cuda.synchronize(dev)
        eager_step = step
        for j in range(ngroups):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for i in range(j * gs, (j + 1) * gs):
                    eager_step(i)
            graphs[j] = g
        eng.use_torch_stream()

        def step(i):                                      # noqa: F811 - replay instead of launching
            if i % gs == 0:                               # the group's graph runs all gs steps; the other calls are no-ops
                graphs[(i // gs) % ngroups].replay()
            return pipe.buffers[i % 2]

    def fence():
        pipe.drain()
        torch.cuda.synchronize(dev)
        if world > 1 or force:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- untimed sanity pass: invariants every cell table must satisfy (no oracle involved) ---------
    step(0)
    fence()
    first_cells = cells_from_torch(cells) if rank == 0 else None
    first_off = slots[(gs - 1) % R][3]                    # (a graph of gs steps leaves the cells of its last step)
    if rank == 0:
        c = first_cells
        ok = ((c["truth_count"] <= c["max_count"]).all() and (c["max_count"] <= N).all()
              and ((c["truth_count"] == c["max_count"]) == (c["hit"] == 1)).all() and (c["n_modes"] >= 1).all())
        if not ok:
            sys.exit("cell table violates its invariants")
    if not no_events:
        eng.drain_kernel_ns()

    # ---- warmup + timed region -------------------------------------------------------------------
    for i in range(args.warmup):
        step(i)
    fence()
    if not no_events:
        eng.drain_kernel_ns()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_timed_in = "the timed region (hipEvents recorded by the library around every launch)"
    # the cell table now holds the LAST TIMED step's chunk (c5: this rank's shard); every rank keeps its own for the oracle
    last_cells = cells_from_torch(cells)
    last_host = None if last is None else last.cpu().numpy().copy()       # the reduced counters of the last TIMED step (the buffers are reused below)
    if split_timing:
        eng_t = Engine(device=local_rank, timing=True)
        eng_t.set_tuning(args.copies, args.threads, args.wg_per_cu, args.unroll)
        nt = max(10, min(args.steps, 100))
        for i in range(nt + 5):
            if i == 5:
                eng_t.sync()
                eng_t.drain_kernel_ns()
            step(args.warmup + args.steps + i, eng_t)
        fence()
        kern_ns, launches = eng_t.drain_kernel_ns()
        eng_t.close()
        kernel_timed_in = f"a second loop of {nt} steps over the same chunks with hipEvent records (the timed region ran without them: they add ~8 us to a ~26 us step)"
    elif no_events:
        kern_ns, launches = int(elapsed * 1e9), args.steps    # no events (inside a graph / --no-timing): wall clock per step (upper bound)
        kernel_timed_in = "nothing: wall clock of the timed region per step (upper bound)"
    else:
        kern_ns, launches = eng.drain_kernel_ns()
    if world > 1 or force:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kmax = torch.tensor([kern_ns / max(launches, 1)], dtype=torch.float64, device=dev)
        kmin = kmax.clone()
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmin, op=dist.ReduceOp.MIN)
        kern_avg_ns = float(kmax.item())
        kern_rank_min_ms, kern_rank_max_ms = float(kmin.item()) / 1e6, float(kmax.item()) / 1e6
        # the exposed all-reduce, measured OUTSIDE the timed region (so that a first multi-GPU run below 6x explains itself: kernel imbalance
        # between the ranks, or the exchange): a few more steps in which the step's stream waits for its collective right behind the vote
        # kernel -- torch events around that wait = from the end of THIS rank's kernel to its buffer holding the sum, the wait for the slowest
        # rank included.  In the timed region the all-reduce of step i runs on RCCL's stream under the kernel of step i + 1.
        exposed_us = None
        if not c5 and not use_graph and args.steps > 0:
            samples = []
            for j in range(6):
                i = args.warmup + args.steps + j
                ans_, tok_, tr_, _ = slots[i % R]
                buf = pipe.acquire(i, zero=True)
                eng.aggregate_device(ans_, tr_, tokens=tok_, counters=buf, cells=cells, cell_tokens=ctok, overwrite=False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
                w.wait()
                e1.record()
                torch.cuda.synchronize(dev)
                if j:
                    samples.append(e0.elapsed_time(e1) * 1e3)
            xm = torch.tensor([sorted(samples)[len(samples) // 2]], dtype=torch.float64, device=dev)
            dist.all_reduce(xm, op=dist.ReduceOp.MAX)
            exposed_us = float(xm.item())
            if not no_events:
                eng.drain_kernel_ns()
    else:
        kern_avg_ns = kern_ns / max(launches, 1)
        kern_rank_min_ms = kern_rank_max_ms = kern_avg_ns / 1e6
        exposed_us = None
    last_slot = slots[(args.warmup + args.steps - 1) % R] if args.steps > 0 else slots[0]

    # ---- the exchange step proves itself (outside the timed region; BASELINE C4 "bit-exact at every GPU count") -----------------
    # Every rank (1) runs its last timed chunk once more into a fresh buffer with NO collective behind it = its PRE-reduce counters,
    # (2) hands them to every other rank over a path that is not the all-reduce under test (a gloo group: pickled numpy over host TCP),
    # (3) sums them in numpy int64 and compares word for word with ITS OWN all-reduced buffer of the last timed step, (4) sends
    # RANK_PARITY_PROBLEMS problems of its last timed chunk through oracle/scv_oracle.c (rank-local, the ranks in parallel), and
    # (5) the verdicts are exchanged so that every rank leaves with the same status; rank 0 names rank and word on a mismatch.
    exchange = None
    if (world > 1 or force) and args.steps > 0:
        ncount = counters_size(B)
        ans, tok, tr, p_off_last = last_slot
        lc = torch.zeros(ncount, dtype=torch.int64, device=dev)
        lt = torch.empty((Pc, B, 16), dtype=torch.uint8, device=dev)
        if Pc:
            eng.aggregate_device(ans, tr, tokens=tok, counters=lc, cells=lt, cell_tokens=ctok, overwrite=False)
        eng.sync()
        torch.cuda.synchronize(dev)
        mine = {"rank": rank, "counters": lc.cpu().numpy(), "cells": lt.cpu().numpy() if c5 else None}
        vg = None if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=vg)
        assert [p_["rank"] for p_ in parts] == list(range(world))
        gathered_table = c5_state["last"].cells.cpu().numpy() if c5 else None
        fault = injected_fault()
        if fault and fault[0] == rank:
            last_host = last_host.copy()
            last_host[fault[1]] += 1
        msgs, want_sum = judge_exchange(rank, ncount, last_host, [p_["counters"] for p_ in parts], gathered_table,
                                        [p_["cells"] for p_ in parts] if c5 else None)
        if first_difference(last_cells.view(np.uint8), mine["cells"] if c5 else lt.cpu().numpy()) is not None:
            msgs.append(f"rank {rank}: the cell table of the last timed step differs from the same chunk's re-run")
        cf_what, cf_msg = closed_form_check(want_sum, B, N, args.problems if c5 else Pc * world, args.dist)
        if cf_msg:
            msgs.append(f"rank {rank}: {cf_msg}")
        n_checked = 0
        if not args.no_cpu_baseline:
            n_checked, msg = check_cells_vs_c_oracle(last_cells, min(RANK_PARITY_PROBLEMS, Pc), B, N, args.seed, args.dist, p_off_last,
                                                     threads=max(1, min(64, (os.cpu_count() or 1) // world)), fatal=False)
            if msg:
                msgs.append(f"rank {rank}: {msg}")
        verdicts = [None] * world
        dist.all_gather_object(verdicts, {"rank": rank, "msgs": msgs, "checked": n_checked}, group=vg)
        failures = [m for v in verdicts for m in v["msgs"]]
        exchange = {
            "allreduce_verified": not failures, "allreduce_words": ncount, "ranks_verified": world,
            "how": "every rank: pre-reduce counters of its last timed chunk (a re-run of the chunk with no collective behind it) -> gloo "
                   "all_gather_object (host TCP, not the all-reduce under test) -> numpy int64 sum == its own all-reduced buffer of the last "
                   "TIMED step, word for word" + ("; C5: its all-gathered cell table == the ranks' own blocks" if c5 else ""),
            "parity_ranks_checked": sum(1 for v in verdicts if v["checked"] > 0),
            "parity_problems_per_rank": min(v["checked"] for v in verdicts),
            "closed_form": cf_what,
        }
        if failures:
            if rank == 0:
                print(json.dumps({"error": "EXCHANGE / PARITY FAILURE", "failures": failures[:8], "n_gpus": world, "value": None,
                                  "metric": "sample-votes/sec (problems x samples)", "allreduce_verified": False}), flush=True)
            sys.exit(3)

    slots_span = f"{min(sl[3] for sl in slots)}..{max(sl[3] for sl in slots) + Pc - 1}" + (" interleaved over the ranks" if world > 1 else "")
    votes_per_step_per_gpu = Pc * B * N
    total_votes = (args.problems * B * N if c5 else votes_per_step_per_gpu * world) * args.steps
    value = total_votes / elapsed
    bytes_per_launch = votes_per_step_per_gpu * BYTES_PER_VOTE * (2 if args.tokens else 1)
    achieved = bytes_per_launch / (kern_avg_ns * 1e-9) / 1e9
    ev = committed_evidence() if (Pc, B, N, args.tokens, args.workload) == (1250, 8, 1 << 20, False, "c3") else {}
    c3_full = full_host = None
    if args.workload == "c3" and world == 1 and not force and args.full_pass and not use_graph and rank == 0:
        c3_full, full_host = full_pass_c3(eng, torch, slots, Pc, B, N, args, cells, ctok)
        if not no_events:
            eng.drain_kernel_ns()
    # the read ceiling of THIS box, measured now (rank 0 of a single-GPU default run; the probe allocates its own 41.9 GB)
    if rank == 0 and world == 1 and ev and not args.no_read_ceiling:
        last_off_kept = last_slot[3]
        slots.clear()                           # the probe allocates its own 41.9 GB: hand the resident chunks back first
        last_slot = (None, None, None, last_off_kept)
        ans = tok = tr = None                   # (the allocation loop's last references)
        torch.cuda.empty_cache()
        live = measure_read_ceiling(chunk_bytes // (4 << 20))
        if live:
            ev["read_ceiling_gbs"] = live
            ev["read_ceiling_source"] = ("tools/hbm_probe.bin --quick run by this process right after the timed region, same box: best of 4 read-only kernels, "
                                         "a LOWER BOUND of the box's read ceiling (the probe's full sweep of the same session finds 1-2 % more: "
                                         "profiles/r04_hbm_probe.log) -- a fraction slightly above 1 of it is probe noise, not a result")
            ev["read_ceiling_is_lower_bound"] = True

    traffic_live = None
    if rank == 0 and world == 1 and not force and ev and args.live_traffic and not use_graph:
        last_off_kept = last_slot[3]
        slots.clear()                           # the child allocates its own chunks: hand the resident ones back first
        last_slot = (None, None, None, last_off_kept)
        ans = tok = tr = None
        torch.cuda.empty_cache()
        t_pm = time.perf_counter()
        traffic_live, traffic_note = measure_traffic_live(args)
        if traffic_live is not None:
            ev["traffic"], ev["traffic_source"] = traffic_live, traffic_note + f" ({time.perf_counter() - t_pm:.0f} s)"
        else:
            WARNINGS.append("roofline.traffic could not be measured in this run (" + str(traffic_note) + "): the committed figure of profiles/ is quoted")
    final = AggregateResult.from_counters(last_host[:counters_size(B)], Pc, B, num_problems=args.problems if c5 else Pc * world)
    if c5:
        workload = (f"C5: P={args.problems} x N={N} sharded by problem over {world} GPU(s) ({Pc} problems = {bytes_per_launch / 1e9:.2f} GB on rank 0); "
                    f"step = vote + counters all-reduce + cell all-gather + {args.resamples}-resample bootstrap (class bound M={c5_state['M']}; "
                    f"with one rank vote and bootstrap are one call, one kernel launch when the shape allows); "
                    f"pass@k sweep k=1..1024 on the host, outside the timed region")
    elif args.workload == "c3":
        workload = (f"C3 streamed in problem-chunks: step = {Pc} problems x {B} budgets x {N} samples int32 "
                    f"({bytes_per_launch / 1e9:.2f} GB) per GPU; 8 steps at 1 GPU = P=10k x B=8 x N=2^20")
    else:
        workload = (f"C2: step = one pass over {Pc} problems x {B} budgets x {N} samples int32 "
                    f"({bytes_per_launch / 1e6:.1f} MB) per GPU, {R} distinct tensors cycled (cold Infinity Cache)")
    out = {
        "metric": "sample-votes/sec (problems x samples)",
        "value": value,
        "unit": "sample-votes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if c5 else "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "distribution": DISTS[args.dist],
            "resident_chunks": R,
            "chunks_distinct": R,
            "chunks_note": (f"{R} distinct 1250-problem chunks of C3's 8 are resident and cycled (global problems "
                            f"{slots_span}); the 8th does not fit: 8 x 41.94 GB = 335.5 GB > HBM") if args.workload == "c3" else None,
            "tokens_stream": bool(args.tokens),
            "parallelism": f"problems sharded over {world} GPU(s), one int64 all-reduce of {counters_size(B)} counters per step",
            "seed": args.seed,
            "launch": (f"hipGraph replay ({gs} step(s) per graph launch)" if use_graph else ("eager, no hipEvent records" if no_events else "eager")),
            "comm": "torch",
            "backend": args.backend if (world > 1 or force) else None,
            "collectives_forced_on_one_rank": bool(force),
            "rccl_ranks": rccl_ranks,
            "hip_devices_visible": ndev,
            "ranks_started_by": ("bench.py itself (torch.distributed.run child)" if os.environ.get("SCV_BENCH_SPAWNED") == "1"
                                 else ("external launcher" if "RANK" in os.environ else None)),
            "hsa_enable_ipc_mode_legacy": dict(HSA_IPC_NOTE),
            "devices_shared_by_ranks": bool(args.share_device),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": ev.get("traffic"),
            "traffic_measured_in_this_run": (traffic_live is not None) if ev.get("traffic") is not None else None,
            "traffic_over_algorithmic": (ev["traffic"] / bytes_per_launch) if ev.get("traffic") else None,
            "traffic_source": ev.get("traffic_source"),
            "kernel": "scv_hist_argmax",
            "kernel_avg_ms": kern_avg_ns / 1e6,
            "kernel_avg_ms_per_rank_min": kern_rank_min_ms,
            "kernel_avg_ms_per_rank_max": kern_rank_max_ms,
            "exposed_allreduce_us": exposed_us,
            "exposed_allreduce_what": ("median of 5 steps AFTER the timed region in which the rank's stream waits for the step's all-reduce right behind its vote kernel "
                                       "(torch events around the wait; max over ranks; the wait for the slowest rank included) -- in the timed region the all-reduce "
                                       "of step i overlaps the kernel of step i + 1") if exposed_us is not None else None,
            "kernel_timed_in": kernel_timed_in,
            "launches_timed": launches,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "measured_read_ceiling_gbs": ev.get("read_ceiling_gbs"),
            # the RAW ratio (round 5 clamped it at 1.0, which would have hidden a wrong byte count or a broken probe): the quick probe is a lower
            # bound of the box's ceiling, so a ratio a little above 1 is probe noise; beyond 1.03 the line says so instead of looking perfect
            "frac_of_measured_read_ceiling": achieved / ev["read_ceiling_gbs"] if ev.get("read_ceiling_gbs") else None,
            "read_ceiling_warning": ("kernel faster than 1.03x the measured pure-read rate: check the probe, the byte count and the timing before quoting this fraction"
                                     if ev.get("read_ceiling_gbs") and achieved / ev["read_ceiling_gbs"] > 1.03 else None),
            "read_ceiling_is_lower_bound": bool(ev.get("read_ceiling_is_lower_bound")),
            "read_ceiling_source": ev.get("read_ceiling_source"),
        },
        "parity": None,
        "parity_ranks_checked": exchange["parity_ranks_checked"] if exchange else None,
        "allreduce_verified": exchange["allreduce_verified"] if exchange else None,
        "exchange_check": exchange,
        "accuracy_last_step": [round(final.accuracy(b), 6) for b in range(B)],
        "counters_sha256_last_step": counters_digest(last_host, counters_size(B)),
        "c3_full": c3_full,
        "device": {"cus": eng.num_cus, "hbm_gb": round(eng.hbm_bytes / 1e9, 1)},
    }
    if rank == 0 and not args.no_cpu_baseline:
        if world == 1:
            out["parity"], out["cpu_baseline"] = cpu_baseline(args, B, N, first_cells, first_off, last_cells, last_slot[3], args.steps > 0)
        else:
            # multi-rank lines carry no CPU baseline (contract: N = 1 only); EVERY rank's shard went through the oracle above
            out["parity"] = (f"bit-exact: EVERY rank's first {exchange['parity_problems_per_rank']} problems x {B} budgets x {N} votes of its last timed "
                             f"chunk ({exchange['parity_ranks_checked']} ranks, rank-local, in parallel) vs oracle/scv_oracle.c; all-reduced counters of "
                             f"every rank == the independently gathered int64 sum ({counters_size(B)} words)") if exchange else None
            out["cpu_baseline"] = None
    elif rank == 0:
        out["cpu_baseline"] = None
    if out["parity"]:
        out["metric"] += ", bit-exact vs CPU (see parity)"
    else:
        out["metric"] += " (parity not checked in this run)"

    # ---- C5 epilogue (outside the timed region): gather the bootstrap slices, host floats, oracle check ----------
    if c5:
        d = c5_state["last"]
        boot_all = passk.gather_bootstrap(d, args.resamples, engine=eng)      # raises on every rank if any rank's device error word is set
        if exchange is not None:
            # the gathered resample table must be the same on every rank (rank 0's goes through the oracle below)
            import hashlib
            digests = [None] * world
            dist.all_gather_object(digests, hashlib.sha256(boot_all.cpu().numpy().tobytes()).hexdigest(), group=vg)
            if len(set(digests)) != 1:
                if rank == 0:
                    print(json.dumps({"error": "EXCHANGE FAILURE: the gathered resample table differs between ranks", "digests": digests,
                                      "n_gpus": world, "value": None, "allreduce_verified": False}), flush=True)
                sys.exit(3)
            exchange["c5_resample_table_equal_on_every_rank"] = True
        if rank == 0:
            all_cells = cells_from_torch(d.cells)
            t1 = time.perf_counter()
            host = passk.finish_host(d.counters.cpu().numpy(), all_cells, boot_all.cpu().numpy(), args.problems, [N])
            out["c5"] = {"accuracy": host["accuracy"], "accuracy_ci95": host["ci95"], "tie_classes_M": d.M,
                         "pass_at_k": {str(k): v[0] for k, v in host["pass_at_k"].items()},
                         "host_float_ms": (time.perf_counter() - t1) * 1e3,
                         "device_pipeline_ms_minus_vote_kernel_ms": elapsed / max(args.steps, 1) * 1e3 - kern_avg_ns / 1e6,
                         "vote_and_bootstrap_in_one_launch": eng.stat("boot_fused") > 0 and eng.stat("boot_separate") == 0,
                         "evaluations_fused_into_one_launch": eng.stat("boot_fused"), "evaluations_as_two_launches": eng.stat("boot_separate"),
                         "semantics": "pass@k and the problem-level bootstrap are NEW (not in the reference): parity unpinned, checked vs the oracle only"}
            if not args.no_cpu_baseline:
                from oracle import coracle
                rc, want_boot = coracle.bootstrap(all_cells, 0, args.resamples, c5_state["boot_seed"], d.M)
                if rc != 0 or not np.array_equal(boot_all.cpu().numpy(), want_boot):
                    sys.exit("PARITY FAILURE: bootstrap table differs from oracle scvo_bootstrap")
                out["c5"]["bootstrap_parity"] = f"all {args.resamples} x {B} x {d.M} resample counters bit-exact vs oracle/scv_oracle.c"
            if args.dump:
                np.savez(args.dump, counters=d.counters.cpu().numpy(), cells=d.cells.cpu().numpy(), boot=boot_all.cpu().numpy())
    elif rank == 0 and args.dump:
        np.savez(args.dump, counters=last_host, **({"full": full_host} if full_host is not None else {}))
    if rank == 0:
        out["warnings"] = list(WARNINGS) or None
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
