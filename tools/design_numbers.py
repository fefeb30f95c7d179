#!/usr/bin/env python3
"""The numbers DESIGN.md / README.md quote, read back from profiles/<tag>_* (one evidence session): print them in the order of the documents'
tables so that the text can be checked against the files.  Usage: design_numbers.py [r06]"""
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
f = lambda n: os.path.join(P, f"{tag}_{n}")  # noqa: E731

d = json.load(open(f("bench.json")))
r = d["roofline"]
print(f"bench: {d['value']:.4e} votes/s, {d['ms_per_step']:.3f} ms/step, kernel {r['kernel_avg_ms']:.3f} ms, {r['achieved'] / 1e3:.3f} TB/s, frac {r['frac']:.3f}, "
      f"quick ceiling {r['measured_read_ceiling_gbs']:.0f} GB/s")
for line in open(f("rocprof_summary.md")):
    if "scv_hist_argmax<" in line and line.startswith("| `void"):
        print("rocprofv3:", " ".join(line.split("|")[2:6]))
print("full probe:", [l.strip() for l in open(f("hbm_probe.log")) if l.startswith("READ_CEILING")])
c = d["cpu_baseline"]
print("reference loop:", c["sample"][:160])
for x in c["dropin_loop"]["results"]:
    print(f"  dropin N={x['N']}: {x['seconds'] * 1e3:.1f} ms = {x['speedup_vs_reference']:.0f}x, engine call {x['split_s']['engine_call'] * 1e6:.0f} us (kernel {x['split_s']['kernel'] * 1e6:.0f}), extract {x['split_s']['extract'] * 1e3:.1f} ms")
fam = c["dropin_loop"]["family"]
print(f"  family: unbatched {fam['dropin_unbatched']['speedup_vs_reference']:.1f}x, {fam['dropin_unbatched']['engine_call_us_mean']:.0f} us/call; batched {fam['dropin_batched']['speedup_vs_reference']:.1f}x, {fam['dropin_batched']['engine_call_us_mean']:.0f} us/call")
print(f"  arithmetic {c['arithmetic']['value']:.2e} / {c['arithmetic']['all_cores']['value']:.2e}; C port {c['c_port']['value']:.2e} / {c['c_port']['all_cores']['value']:.2e}")
for name in ("bench_c5_1gpu", "bench_c2", "bench_c2_graph", "bench_c2_graph10", "bench_comm_peer_2ctx", "bench_comm_rccl_1gpu"):
    x = json.load(open(f(name + ".json")))
    extra = x["roofline"].get("exposed_allreduce_us")
    print(f"{name}: {x['value']:.4e} votes/s, {x['ms_per_step'] * 1e3:.2f} us/step" + (f", all-reduce exposed {extra:.0f} us" if extra else ""))
for l in open(f("bench_dists.jsonl")):
    x = json.loads(l); print("dist", x["config"]["distribution"], round(x["roofline"]["achieved"]), "GB/s")
for l in open(f("bench_tokens.jsonl")):
    x = json.loads(l); print("tokens", round(x["roofline"]["achieved"]), "GB/s")
print("--- regimes (TB/s)")
for l in open(f("regimes.log")):
    m = re.match(r"(.*?)\s+\[(\d+), (\d+), (\d+)\]\s+tok=(\d)\s+([\d.]+) us\s+([\d.]+) GB/s", l)
    if m:
        print(f"{m.group(1).strip():44s} {float(m.group(7)) / 1e3:5.2f} TB/s {float(m.group(6)):8.1f} us")
    elif "prefix one-pass" in l:
        print(l.strip()[:110])
for name in ("prefix_small", "prefix_small_promised", "prefix_small_lane"):
    print("---", name)
    for l in open(f(name + ".log")):
        if l.startswith("{"):
            x = json.loads(l); print(f"  {x['shape']} tokens={x['tokens']}: {x['prefix_us']:.1f} us (dense {x.get('dense_us', float('nan')):.1f})")
for name in ("prefix_dists.log", "crossovers.md", "packed_records.log"):
    if os.path.exists(f(name)):
        print("---", name); print(open(f(name)).read().rstrip())
r = d["roofline"]
print(f"traffic: {r['traffic']} B measured in this run: {r['traffic_measured_in_this_run']}, over algorithmic {r['traffic_over_algorithmic']}")
print("c3_full:", json.dumps(d.get("c3_full"))[:400])
rows = [json.loads(l) for l in open(f("rtn_ab.log")) if l.startswith("{")]
for N in sorted({x["N"] for x in rows}):
    rr = [x for x in rows if x["N"] == N]
    print(f"rtn A/B N={N}: reg {min(x['reg_cells_TBps'] for x in rr):.1f}-{max(x['reg_cells_TBps'] for x in rr):.1f}  rtn16 {min(x['rtn_g16_TBps'] for x in rr):.1f}-{max(x['rtn_g16_TBps'] for x in rr):.1f}")
print("pytest:", [l.strip() for l in open(f("pytest_gpu.log")) if "passed" in l])
