"""bench.py's command line where no GPU is needed: the multi-GPU form of the driver's command on a box with fewer
devices than ranks (here: none) must answer with ONE JSON line carrying "error" and a non-zero exit status -- never a
traceback, never silence -- and must not need a launcher in front."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_launcher_and_without_devices_is_one_error_line():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered by the gpu-marked form (device count + 1 ranks)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO, env=env)
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    d = json.loads(lines[0])
    assert "error" in d and "--gpus 8" in d["error"] and d["n_gpus"] == 8 and d["hip_devices_visible"] == 0


def test_spawn_command_is_the_drivers_own_launcher_line(monkeypatch):
    """What bench.py runs for `python bench.py --gpus 4 --steps 2`: torch.distributed.run, one node, 4 processes,
    rendezvous on 127.0.0.1, the same script with the same arguments (the contract's launch line)."""
    import importlib
    import types
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env", {})
        return types.SimpleNamespace(returncode=0, stdout='{"metric": "m", "n_gpus": 4}\n')

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--share-device", "--backend", "gloo"])
    args = bench.parse()
    assert bench.spawn_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "2", "--share-device", "--backend", "gloo"][-7:]
    assert seen["env"]["SCV_BENCH_SPAWNED"] == "1"
