"""Build recipe for the HIP extension: hipcc cross-compiles gfx950 without a GPU.

The ~120 kernel instantiations are spread over several translation units (csrc/scvote_dispatch.h); each is compiled
to an object file under csrc/build/ by its own hipcc process (in parallel, skipped when the object is newer than its
sources) and the objects are linked into csrc/libscvote.so, in-tree, so that it travels to the GPU box."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(CSRC, "libscvote.so")
UNITS = ["scvote.hip", "scvote_stream_c4.hip", "scvote_stream_c8.hip", "scvote_stream_c16.hip",
         "scvote_reg_g8.hip", "scvote_reg_g16.hip", "scvote_reg_g32.hip", "scvote_reg_g64.hip", "scvote_dense.hip", "scvote_comm.hip", "scvote_sort.hip", "scvote_prefix.hip", "scvote_sort_prefix.hip"]
HEADERS = [os.path.join(CSRC, "scvote_kernels.hip.h"), os.path.join(CSRC, "scvote_dispatch.h"), os.path.join(CSRC, "scvote_hostpool.h"),
           os.path.join(os.path.dirname(HERE), "include", "scvote.h")]
UNIT_HEADERS = {"scvote_sort.hip": [os.path.join(CSRC, "scvote_sort.hip.h"), os.path.join(CSRC, "scvote_sortnet.h")],
                "scvote_prefix.hip": [os.path.join(CSRC, "scvote_prefix.hip.h")],
                "scvote_sort_prefix.hip": [os.path.join(CSRC, "scvote_sort_prefix.hip.h"), os.path.join(CSRC, "scvote_sort.hip.h"), os.path.join(CSRC, "scvote_sortnet.h")]}      # headers only one unit includes
SOURCES = UNITS + ["scvote_hostpool.h", "scvote_kernels.hip.h", "scvote_sort.hip.h", "scvote_sortnet.h", "scvote_prefix.hip.h", "scvote_sort_prefix.hip.h", "scvote_dispatch.h"]          # (tools/kernel_resources.py lists them)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc() -> str:
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _obj(unit: str) -> str:
    return os.path.join(OBJDIR, unit.replace(".hip", ".o"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB_PATH, [os.path.join(CSRC, u) for u in UNITS] + HEADERS + [h for hs in UNIT_HEADERS.values() for h in hs])


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """hipcc --offload-arch=gfx950: one object per translation unit (parallel) -> csrc/libscvote.so."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJDIR, exist_ok=True)
    todo = [u for u in UNITS if force or _stale(_obj(u), [os.path.join(CSRC, u)] + HEADERS + UNIT_HEADERS.get(u, []))]

    def compile_unit(unit):
        cmd = [_hipcc(), *FLAGS, "-c", "-o", _obj(unit), os.path.join(CSRC, unit)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 4)) as pool:
        list(pool.map(compile_unit, todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *[_obj(u) for u in UNITS], "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


# ---- variants of the library that are NOT the product (tests and tools load them through SCV_LIB_PATH) ---------------------------
#   hooks  the fault-injection hooks (SCV_TEST_FAULT) compiled in: -DSCV_TEST_HOOKS on the two host-side units; every kernel object is the
#          product's own.  tests/test_gpu_parity.py::test_no_cpp_exception_crosses_the_abi loads it; libscvote.so never reads the environment.
#   tsan   the same, host code under -fsanitize=thread (device code unchanged: the flag is ignored for amdgcn).  Best effort on a GPU box
#          (tools/tsan_host.sh): the HIP runtime itself is not TSAN-clean; the threads of the library are checked on the CPU instead
#          (tests/hostpool_sanitize.cpp over csrc/scvote_hostpool.h).
VARIANTS = {"hooks": (["-DSCV_TEST_HOOKS"], []),
            "tsan": (["-DSCV_TEST_HOOKS", "-g", "-Xarch_host", "-fsanitize=thread"], ["-fsanitize=thread"])}
VARIANT_UNITS = ["scvote.hip", "scvote_comm.hip"]          # the host-side units; the other objects are shared with the product


def variant_path(name: str) -> str:
    return os.path.join(CSRC, f"libscvote_{name}.so")


def build_variant(name: str, force: bool = False, verbose: bool = False) -> str:
    cflags, lflags = VARIANTS[name]
    build(force=False, verbose=verbose)                    # the shared objects
    out = variant_path(name)
    vdir = os.path.join(OBJDIR, name)
    os.makedirs(vdir, exist_ok=True)

    def vobj(unit):
        return os.path.join(vdir, unit.replace(".hip", ".o"))

    todo = [u for u in VARIANT_UNITS if force or _stale(vobj(u), [os.path.join(CSRC, u)] + HEADERS + UNIT_HEADERS.get(u, []))]

    def compile_unit(unit):
        cmd = [_hipcc(), *FLAGS, *cflags, "-Wno-option-ignored", "-c", "-o", vobj(unit), os.path.join(CSRC, unit)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=2) as pool:
        list(pool.map(compile_unit, todo))
    objs = [vobj(u) if u in VARIANT_UNITS else _obj(u) for u in UNITS]
    if todo or force or _stale(out, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *lflags, "-o", out, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True))
    for v in sys.argv[1:]:
        print(build_variant(v, force=True, verbose=True))
