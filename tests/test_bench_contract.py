"""bench.py contract on the GPU box: one JSON line with the driver's keys (+ roofline, cpu_baseline), and
the N > 1 control flow (two ranks, gloo, both on cuda:0 -- everything except RCCL itself)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
SMALL = ["--problems-per-step", "48", "--samples", str(1 << 16), "--steps", "3", "--warmup", "1"]


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *SMALL, "--cpu-baseline-seconds", "0.5"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "int32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["parity"].startswith("bit-exact")
    assert abs(d["value"] - 48 * 8 * (1 << 16) * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9


def test_two_ranks_share_the_gpu_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", *SMALL,
           "--backend", "gloo", "--share-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and d["parity"] is None
    single = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *SMALL, "--no-cpu-baseline"],
                            capture_output=True, text=True, timeout=600, cwd=REPO)
    one = _last_json(single.stdout)
    # weak scaling: rank r streams its own chunks, accuracy is over all problems of the last step --
    # same generator, so both runs sit at the same accuracy to within sampling of different problems
    assert all(0.0 <= a <= 1.0 for a in d["accuracy_last_step"]) and len(d["accuracy_last_step"]) == 8
    assert abs(d["accuracy_last_step"][0] - one["accuracy_last_step"][0]) < 0.2
