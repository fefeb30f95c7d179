"""ctypes binding of include/scvote.h.  Fails loudly when the HIP library is missing."""
from __future__ import annotations

import ctypes as C
import os

from ._build import LIB_PATH

_lib = None

# include/scvote.h constants
MEM_HOST, MEM_DEVICE = 0, 1
FLAG_TIMING, FLAG_CLAMP, FLAG_PACKED_CELLS = 0x1, 0x2, 0x4
COMM_PEER, COMM_RCCL = 0x0, 0x1
DIST_UNIFORM, DIST_PEAKED, DIST_DEGENERATE, DIST_TIE, DIST_PEAKED_WRONG, DIST_DEGENERATE_WRONG = 0, 1, 2, 3, 4, 5
NUM_BINS, TIE_CLASSES = 1024, 1025
OK, ERR_ARG, ERR_DOMAIN, ERR_NO_DEVICE, ERR_NOT_TIMED, ERR_ALLOC = 0, -2001, -2002, -2003, -2004, -2005


class ScvError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"scvote error {code}: {message}")
        self.code = code


class DomainError(ScvError, ValueError):
    """A vote outside bins 0..1023 reached the engine (SCV_ERR_DOMAIN)."""


def load():
    """dlopen csrc/libscvote.so and declare every entry point of include/scvote.h."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = os.environ.get("SCV_LIB_PATH") or LIB_PATH          # (A/B runs of two builds of the library; tools/ only)
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    # One HIP runtime per process.  Device pointers and stream handles cross this boundary from
    # torch, so libscvote must bind to the libamdhip64 torch has loaded (same SONAME, resolved to the
    # already-loaded copy); loading ours first would bring in a second runtime from /opt/rocm.
    import torch  # noqa: F401
    L = C.CDLL(lib_path)
    p, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    L.scv_create.argtypes = [C.POINTER(p), C.c_int, u32]
    L.scv_destroy.argtypes = [p]
    L.scv_set_stream.argtypes = [p, p]
    L.scv_sync.argtypes = [p]
    L.scv_set_tuning.argtypes = [p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.scv_set_option.argtypes = [p, C.c_char_p, i64]
    L.scv_aggregate_i32.argtypes = [p, p, p, p, p, i64, i32, i64, C.c_int, p, p, p, p, p]
    L.scv_aggregate_prefix_i32.argtypes = [p, p, p, p, p, i64, i32, i64, C.c_int, p, p, p, p, p]
    L.scv_bootstrap.argtypes = [p, p, i64, i32, i32, i32, u64, i32, C.c_int, p]
    L.scv_aggregate_bootstrap_i32.argtypes = [p, p, p, p, p, i64, i32, i64, p, p, p, p, p, i32, i32, u64, i32, p]
    L.scv_synth_fill_i32.argtypes = [p, p, p, p, i64, i32, i64, i64, u64, C.c_int]
    L.scv_export_error_word.argtypes = [p, p]
    L.scv_comm_create.argtypes = [C.POINTER(p), C.POINTER(C.c_int), C.c_int, u32, u32]
    L.scv_comm_destroy.argtypes = [p]
    L.scv_comm_size.argtypes = [p]
    L.scv_comm_ctx.argtypes = [p, C.c_int]
    L.scv_comm_ctx.restype = p
    L.scv_allreduce_counters.argtypes = [p, C.POINTER(p), i64]
    L.scv_comm_sync.argtypes = [p]
    ab_build = bool(os.environ.get("SCV_LIB_PATH"))              # an older build loaded for an A/B run (tools/ only) may lack the newest entry points
    if not ab_build or hasattr(L, "scv_allgather_cells"):
        L.scv_allgather_cells.argtypes = [p, C.POINTER(p), C.POINTER(i64), i32]
        L.scv_allgather_i64.argtypes = [p, C.POINTER(p), C.POINTER(i64)]
        L.scv_comm_get_stat.argtypes = [p, C.c_char_p, C.POINTER(i64)]
    L.scv_last_kernel_ns.argtypes = [p, C.POINTER(u64)]
    L.scv_drain_kernel_ns.argtypes = [p, C.POINTER(u64), C.POINTER(u64)]
    L.scv_get_stat.argtypes = [p, C.c_char_p, C.POINTER(i64)]
    L.scv_host_alloc.argtypes = [C.POINTER(p), C.c_size_t]
    L.scv_host_free.argtypes = [p]
    L.scv_device_count.argtypes = []
    L.scv_device_info.argtypes = [p, C.POINTER(i64 * 4)]
    L.scv_last_error.argtypes = []
    L.scv_last_error.restype = C.c_char_p
    L.scv_version.argtypes = []
    L.scv_version.restype = C.c_char_p
    for name in ("scv_create", "scv_destroy", "scv_set_stream", "scv_sync", "scv_set_tuning", "scv_set_option", "scv_aggregate_i32", "scv_aggregate_prefix_i32",
                 "scv_bootstrap", "scv_aggregate_bootstrap_i32", "scv_synth_fill_i32", "scv_last_kernel_ns", "scv_drain_kernel_ns",
                 "scv_device_count", "scv_device_info", "scv_host_alloc", "scv_host_free", "scv_get_stat", "scv_export_error_word",
                 "scv_comm_create", "scv_comm_destroy", "scv_comm_size", "scv_allreduce_counters", "scv_comm_sync",
                 "scv_allgather_cells", "scv_allgather_i64", "scv_comm_get_stat"):
        if ab_build and not hasattr(L, name):
            continue
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc: int):
    if rc == OK:
        return
    msg = load().scv_last_error().decode("utf-8", "replace")
    if rc == ERR_DOMAIN:
        raise DomainError(rc, msg)
    raise ScvError(rc, msg)
