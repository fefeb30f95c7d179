"""Build recipe for the HIP extension: hipcc cross-compiles gfx950 without a GPU (seconds)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libscvote.so")
SOURCES = ["scvote.hip", "scvote_kernels.hip.h"]
HEADER = os.path.join(os.path.dirname(HERE), "include", "scvote.h")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> csrc/libscvote.so (in-tree, so it travels to the GPU box)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB_PATH, os.path.join(CSRC, "scvote.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
