#!/usr/bin/env python3
"""Docs hygiene: every evidence file a document cites exists, and the documents that describe the CURRENT state cite
the CURRENT round's evidence.

  * DESIGN.md, INTEGRATION.md, README.md, profiles/README.md: every `profiles/rNN_*` / `rNN_*.{json,jsonl,log,md,csv}` they name must exist under
    profiles/ (brace lists `r06_bench_c2{,_graph}.json` and `*` globs are expanded).
  * INTEGRATION.md and README.md may cite only the current round's files (the round is read from DESIGN.md's title).
  * DESIGN.md may cite older rounds only for the measurements listed in OLD_OK -- probes and A/B logs that were run once and are not repeated
    every round; each entry says why.
Exit status 1 with the list of offences; tests/test_docs.py runs it.  Usage: python tools/check_docs.py"""
import glob
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")

OLD_OK = {
    "r02_fetch_size_calibration.json": "the x2 of FETCH_SIZE on gfx950, calibrated once on a known-bytes read kernel",
    "r04_valu_probe.log": "VALU issue rates of gfx950 (a property of the part, probed once)",
    "r04_c2_sweep.json": "launch forms of C2 (eager / graph / one launch), swept once",
    "r01_crossover_r4_d1.log": "the first measurement of fused atomics against the reduction (superseded by r06_crossovers.md, kept as history)",
    "r05_rtn_ab.log": "ranks from returning atomics against the register-resident kernels: the A/B of round 5, not repeated",
    "r05_rtn_ranks_ab.log": "the same before the data pivot",
    "r05_sort_prefix_timeline.log": "phase timeline of scv_sort_prefix (measurement build), round 5",
    "r05_sort_prefix_timeline_direct_stores.log": "the same with records written straight from the lanes",
    "r05_sort_prefix_scaling.log": "scv_sort_prefix launch time against the number of pools, round 5",
    "r05_prefix_latency.log": "reference-sized prefix calls (P = 30 .. 3000), round 5",
    "r05_prefix_pool_v1_ab.log": "scv_prefix_pool v1 (64 lanes per problem) A/B, round 5",
    "r05_hbm_probe_percu.log": "per-CU read rate of the part (pure-read probe by workgroup count)",
    "r05_regimes_pmc.md": "round 5's PMC of the 128-slot shape at N = 96 (the shape round 6 replaced: the before of the A/B)",
    "r05_regimes.log": "round 5's regimes table (the before of round 6's A/B rows)",
    "r05_prefix_small.log": "round 5's prefix timings (the before column)",
    "r05_prefix_pmc.md": "round 5's PMC of the prefix kernels (unchanged kernels)",
}


def expand(token):
    """`r06_bench_c2{,_graph}.json` -> both names; `*` is left to glob."""
    m = re.search(r"\{([^{}]*)\}", token)
    if not m:
        return [token]
    out = []
    for alt in m.group(1).split(","):
        out += expand(token[:m.start()] + alt + token[m.end():])
    return out


def cited(text):
    toks = set()
    for m in re.finditer(r"(?:profiles/)?(r\d\d_[A-Za-z0-9_{},*.\-]*?\.(?:jsonl|json|log|md|csv))", text):
        toks.add(m.group(1))
    return toks


def main():
    design = open(os.path.join(R, "DESIGN.md")).read()
    m = re.search(r"round (\d+)", design.splitlines()[0])
    cur = f"r{int(m.group(1)):02d}" if m else None
    bad = []
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md", os.path.join("profiles", "README.md")):
        path = os.path.join(R, doc)
        if not os.path.exists(path):
            continue
        for tok in sorted(cited(open(path).read())):
            for name in expand(tok):
                hits = glob.glob(os.path.join(P, name))
                if not hits:
                    bad.append(f"{doc}: cites {name}, which is not under profiles/")
                    continue
                rnd = name[:3]
                if doc.startswith("profiles"):
                    continue                                   # the index of every round's files
                if cur and rnd != cur:
                    if doc in ("INTEGRATION.md", "README.md"):
                        bad.append(f"{doc}: cites {name} (round {rnd[1:]}), but the current state is round {cur[1:]}: quote {cur}_* or drop the number")
                    elif not any(os.path.basename(h) in OLD_OK for h in hits):
                        bad.append(f"{doc}: cites {name} of an older round without an entry in tools/check_docs.py OLD_OK (why is it still the evidence?)")
    for b in bad:
        print(b)
    print(f"check_docs: current round {cur}; {len(bad)} offence(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
