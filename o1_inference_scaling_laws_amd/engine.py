"""Python face of the HIP engine (ctypes over include/scvote.h).

Two calling modes, mirroring ``mem_kind`` of the ABI:

* numpy arrays  -> SCV_MEM_HOST: the library stages problem-chunks through HBM and returns numpy.
* torch CUDA tensors -> SCV_MEM_DEVICE: zero-copy, asynchronous on torch's current stream; outputs
  are torch tensors on the same device (PyTorch is used for device memory and streams only).

There is no CPU implementation here.  Without the compiled library or a HIP device, constructing
an ``Engine`` raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import TIE_CLASSES, check
from .scoring import accuracy_from_tie_classes, avg_tokens_used, exact_accuracy_from_tie_classes

CELL_DTYPE = np.dtype(
    [("max_count", "<u4"), ("truth_count", "<u4"), ("n_modes", "<u2"), ("min_mode", "<i2"),
     ("hit", "u1"), ("pad", "u1", (3,))]
)
assert CELL_DTYPE.itemsize == 16


def counters_size(B: int) -> int:
    """int64 words of the packed per-budget counters: tie_class_hits[B,1025] | token_sum[B] | truth_count_sum[B]."""
    return B * TIE_CLASSES + 2 * B


@dataclass
class AggregateResult:
    """Integer outputs of one aggregation (host copies) + the reference's floats derived from them."""
    P: int
    B: int
    cells: np.ndarray | None            # CELL_DTYPE [P, B]
    cell_tokens: np.ndarray | None      # int64 [P, B]
    tie_class_hits: np.ndarray          # int64 [B, 1025]
    token_sum: np.ndarray               # int64 [B]
    truth_count_sum: np.ndarray         # int64 [B]
    num_problems: int | None = None     # denominator for accuracy (global P when sharded)

    def _den(self):
        return self.num_problems if self.num_problems is not None else self.P

    def accuracy(self, b: int = 0) -> float:
        return accuracy_from_tie_classes(self.tie_class_hits[b], self._den())

    def exact_accuracy(self, b: int = 0):
        return exact_accuracy_from_tie_classes(self.tie_class_hits[b], self._den())

    def avg_tokens_used(self, b: int = 0) -> np.float64:
        return avg_tokens_used(self.token_sum[b], self._den())

    @staticmethod
    def from_counters(counters: np.ndarray, P: int, B: int, cells=None, cell_tokens=None, num_problems=None):
        counters = np.asarray(counters, dtype=np.int64)
        tie = counters[: B * TIE_CLASSES].reshape(B, TIE_CLASSES)
        tok = counters[B * TIE_CLASSES: B * TIE_CLASSES + B]
        tcs = counters[B * TIE_CLASSES + B:]
        return AggregateResult(P, B, cells, cell_tokens, tie, tok, tcs, num_problems)


def _np_ptr(a):
    # (the address as a plain int: ndarray.ctypes.data_as(c_void_p) costs 2.2 us per argument, a fifth of a reference-sized call)
    return None if a is None else a.ctypes.data


def _out(shape, dtype, written: bool):
    """Output array of a HOST-mode call: the library overwrites every element when the call has work (P, B > 0)."""
    return np.empty(shape, dtype=dtype) if written else np.zeros(shape, dtype=dtype)


def pinned_empty(shape, dtype=np.int32) -> np.ndarray:
    """numpy array over page-locked host memory (scv_host_alloc): HOST-mode calls DMA such inputs in place instead
    of copying them through the pinned bounce slots first.  The memory is released when the array is collected."""
    import weakref
    L = _lib.load()
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes == 0:
        return np.empty(shape, dtype=dtype)
    ptr = C.c_void_p()
    check(L.scv_host_alloc(C.byref(ptr), nbytes))
    buf = (C.c_char * nbytes).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    weakref.finalize(buf, L.scv_host_free, C.c_void_p(ptr.value))
    return arr


class Engine:
    """One context per GPU per process (one process per GPU under torch.distributed)."""

    def __init__(self, device: int | None = None, timing: bool = False, clamp_to_invalid_bin: bool = False, packed_cells: bool = False, _adopt=None):
        """``packed_cells=True`` (SCV_FLAG_PACKED_CELLS, opt-in): ``aggregate_device`` over cells of up to 127 votes writes 4-byte records --
        its ``cells`` tensor is uint8 [P, B, 4] and ``cells_from_torch`` decodes it (``unpack_cells``); every other call form of such an engine
        must be made without a cell table (``cells=False``)."""
        self._L = _lib.load()            # raises ImportError when csrc/libscvote.so is missing
        self._ctx = C.c_void_p()
        self.packed_cells = bool(packed_cells)
        flags = (_lib.FLAG_TIMING if timing else 0) | (_lib.FLAG_CLAMP if clamp_to_invalid_bin else 0) | (_lib.FLAG_PACKED_CELLS if packed_cells else 0)
        if device is None:
            import torch
            device = torch.cuda.current_device() if torch.cuda.is_available() else -1
        self.device = int(device)
        self._owns_ctx = _adopt is None
        if _adopt is None:
            check(self._L.scv_create(C.byref(self._ctx), self.device, flags))
        else:                            # a context owned by a communicator (scv_comm_ctx): never destroyed from here
            self._ctx = C.c_void_p(_adopt)
        info = (C.c_int64 * 4)()
        check(self._L.scv_device_info(self._ctx, C.byref(info)))
        self.num_cus, self.lds_bytes, self.clock_khz, self.hbm_bytes = (int(x) for x in info)
        self._bound_stream = None
        self._overwrite = False
        self._options = {}
        self.timing = timing

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            if self._owns_ctx:
                self._L.scv_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- configuration ----------------------------------------------------------------------

    def set_tuning(self, copies: int = 0, threads: int = 0, wg_per_cu: int = 0, unroll: int = 0):
        check(self._L.scv_set_tuning(self._ctx, copies, threads, wg_per_cu, unroll))

    def set_option(self, key: str, value: int):
        check(self._L.scv_set_option(self._ctx, key.encode(), int(value)))
        self._options[key] = int(value)

    @staticmethod
    def budgets_come_out_of_one_sort(budgets, N: int) -> bool:
        """True when a DEVICE-mode prefix call over pools of N votes with these budgets may PROMISE them to the library
        (option "prefix_path" = 5, include/scvote.h): pools of 17 .. 128 votes (N % 4 == 0) whose budgets are all 0, a power of
        two <= 16 / 32 / 64 (pools <= 32 / 64 / 128) or >= N -- the reference's own lists (o1.py:274-277: 1, 2, 4 ... N)."""
        if not (16 < N <= 128) or N % 4:
            return False
        cap = 16 if N <= 32 else (32 if N <= 64 else 64)
        return all(n <= 0 or n >= N or ((n & (n - 1)) == 0 and n <= cap) for n in (int(x) for x in budgets))

    def _check_device(self, tensor, name="tensor"):
        if self.device >= 0 and tensor.device.index != self.device:
            raise ValueError(f"{name} lives on {tensor.device} but this engine is bound to cuda:{self.device}")

    def use_torch_stream(self):
        """Launch on torch's current stream OF THE ENGINE'S DEVICE so engine work orders with torch / RCCL ops."""
        import torch
        dev = self.device if self.device >= 0 else None
        s = int(torch.cuda.current_stream(dev).cuda_stream)   # 0 = the default stream: borrowed as such
        if s != self._bound_stream:
            check(self._L.scv_set_stream(self._ctx, C.c_void_p(s)))
            self._bound_stream = s

    def sync(self):
        check(self._L.scv_sync(self._ctx))

    @staticmethod
    def pinned_empty(shape, dtype=np.int32) -> np.ndarray:
        """Page-locked host memory for HOST-mode inputs (module-level ``pinned_empty``; scv_host_alloc)."""
        return pinned_empty(shape, dtype)

    def stat(self, key: str) -> int:
        """Monotonic counters of the single-launch forms taken by this context (scv_get_stat)."""
        v = C.c_int64()
        check(self._L.scv_get_stat(self._ctx, key.encode(), C.byref(v)))
        return int(v.value)

    def drain_kernel_ns(self):
        """(total_ns, launches) of the timed hot-path launches since the previous drain."""
        tot, n = C.c_uint64(), C.c_uint64()
        check(self._L.scv_drain_kernel_ns(self._ctx, C.byref(tot), C.byref(n)))
        return int(tot.value), int(n.value)

    # ---- HOST mode ------------------------------------------------------------------------------

    def aggregate(self, answers, truth, tokens=None, n_valid=None, want_cells=True) -> AggregateResult:
        """answers int32 [P,B,N] (numpy) -> AggregateResult.  Blocking.  See scv_aggregate_i32."""
        answers = np.ascontiguousarray(answers, dtype=np.int32)
        if answers.ndim != 3:
            raise ValueError("answers must be [P, B, N]")
        P, B, N = answers.shape
        truth = np.ascontiguousarray(truth, dtype=np.int32)
        if truth.shape != (P,):
            raise ValueError("truth must be [P]")
        if tokens is not None:
            tokens = np.ascontiguousarray(tokens, dtype=np.int32)
            if tokens.shape != answers.shape:
                raise ValueError("tokens must match answers")
        if n_valid is not None:
            n_valid = np.ascontiguousarray(n_valid, dtype=np.int32)
            if n_valid.shape != (B,):
                raise ValueError("n_valid must be [B]")
        w = P > 0 and B > 0
        cells = _out((P, B), CELL_DTYPE, w) if want_cells else None
        cell_tokens = _out((P, B), np.int64, w) if (want_cells and tokens is not None) else None
        tie = _out((B, TIE_CLASSES), np.int64, w)
        tok = _out((B,), np.int64, w and tokens is not None)
        tcs = _out((B,), np.int64, w)
        check(self._L.scv_aggregate_i32(self._ctx, _np_ptr(answers), _np_ptr(tokens), _np_ptr(n_valid),
                                        _np_ptr(truth), P, B, N, _lib.MEM_HOST, _np_ptr(cells),
                                        _np_ptr(cell_tokens), _np_ptr(tie), _np_ptr(tok), _np_ptr(tcs)))
        return AggregateResult(P, B, cells, cell_tokens, tie, tok, tcs)

    def aggregate_prefix(self, pool, truth, n_valid, tokens=None, want_cells=True) -> AggregateResult:
        """pool int32 [P,N] (numpy), n_valid int32 [B]: budget b votes over pool[p, :n_valid[b]].
        One pass over the pool (scv_aggregate_prefix_i32); same result as ``aggregate`` on the dense
        expansion.  Blocking."""
        pool = np.ascontiguousarray(pool, dtype=np.int32)
        if pool.ndim != 2:
            raise ValueError("pool must be [P, N]")
        P, N = pool.shape
        truth = np.ascontiguousarray(truth, dtype=np.int32)
        n_valid = np.ascontiguousarray(n_valid, dtype=np.int32)
        if truth.shape != (P,) or n_valid.ndim != 1:
            raise ValueError("truth must be [P] and n_valid [B]")
        B = n_valid.shape[0]
        if tokens is not None:
            tokens = np.ascontiguousarray(tokens, dtype=np.int32)
            if tokens.shape != pool.shape:
                raise ValueError("tokens must match pool")
        w = P > 0 and B > 0
        cells = _out((P, B), CELL_DTYPE, w) if want_cells else None
        cell_tokens = _out((P, B), np.int64, w) if (want_cells and tokens is not None) else None
        tie = _out((B, TIE_CLASSES), np.int64, w)
        tok = _out((B,), np.int64, w and tokens is not None)
        tcs = _out((B,), np.int64, w)
        check(self._L.scv_aggregate_prefix_i32(self._ctx, _np_ptr(pool), _np_ptr(tokens), _np_ptr(n_valid),
                                               _np_ptr(truth), P, B, N, _lib.MEM_HOST, _np_ptr(cells),
                                               _np_ptr(cell_tokens), _np_ptr(tie), _np_ptr(tok), _np_ptr(tcs)))
        return AggregateResult(P, B, cells, cell_tokens, tie, tok, tcs)

    def bootstrap(self, cells: np.ndarray, r_begin: int, r_end: int, seed: int, M: int) -> np.ndarray:
        """cells CELL_DTYPE [P,B] -> int64 [r_end-r_begin, B, M].  Blocking.  See scv_bootstrap."""
        cells = np.ascontiguousarray(cells)
        if cells.dtype != CELL_DTYPE or cells.ndim != 2:
            raise ValueError("cells must be CELL_DTYPE [P, B]")
        P, B = cells.shape
        out = np.zeros((r_end - r_begin, B, M), dtype=np.int64)
        check(self._L.scv_bootstrap(self._ctx, _np_ptr(cells), P, B, r_begin, r_end, seed, M, _lib.MEM_HOST, _np_ptr(out)))
        return out

    # ---- DEVICE mode (torch tensors; asynchronous on torch's current stream) --------------------

    def _device_call(self, votes, votes_name, row_shape, truth, tokens, n_valid, counters, cells, cell_tokens, overwrite, want_no_cells=True):
        """Shared by the three DEVICE-mode entry points: validates the tensors (contiguous CUDA int32 of the right shape on the
        engine's device), binds torch's current stream, allocates the outputs the caller did not pass (``cells=False``: none) and
        sets the overwrite option.  Returns (P, B, N, counters, cells, cell_tokens, pointer-of, counter pointers)."""
        import torch
        nd = len(row_shape) + 1
        if not (votes.is_cuda and votes.dtype == torch.int32 and votes.is_contiguous() and votes.dim() == nd):
            raise ValueError(f"{votes_name} must be a contiguous CUDA int32 tensor [{'P, B, N' if nd == 3 else 'P, N'}]")
        P, N = int(votes.shape[0]), int(votes.shape[-1])
        B = int(votes.shape[1]) if nd == 3 else int(n_valid.shape[0])
        dev = votes.device
        self._check_device(votes, votes_name)
        if truth is None:
            raise ValueError("truth is required")
        for name, t, shape in (("truth", truth, (P,)), ("tokens", tokens, tuple(votes.shape)), ("n_valid", n_valid, (B,))):
            if t is None:
                continue
            if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and tuple(t.shape) == shape and t.device == dev):
                raise ValueError(f"{name} must be a contiguous CUDA int32 tensor {shape} on {dev}")
        self.use_torch_stream()
        if counters is None:
            counters = torch.zeros(counters_size(B), dtype=torch.int64, device=dev)
        elif not (counters.is_cuda and counters.dtype == torch.int64 and counters.numel() == counters_size(B) and counters.is_contiguous()):
            raise ValueError("counters must be int64 [counters_size(B)] on the device")
        if cells is None:
            cells = torch.empty((P, B, 4 if getattr(self, "packed_cells", False) else 16), dtype=torch.uint8, device=dev)
        elif cells is False:
            if not want_no_cells:
                raise ValueError("this call needs the cell table")
            cells = None
        if cell_tokens is None and tokens is not None and cells is not None:
            cell_tokens = torch.empty((P, B), dtype=torch.int64, device=dev)
        if overwrite != self._overwrite:
            check(self._L.scv_set_option(self._ctx, b"overwrite_counters", int(bool(overwrite))))
            self._overwrite = bool(overwrite)
        base = counters.data_ptr()
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        cptrs = (C.c_void_p(base), C.c_void_p(base + 8 * B * TIE_CLASSES), C.c_void_p(base + 8 * (B * TIE_CLASSES + B)))
        return P, B, N, counters, cells, cell_tokens, ptr, cptrs

    def aggregate_device(self, answers, truth, tokens=None, n_valid=None, counters=None, cells=None,
                         cell_tokens=None, overwrite=False):
        """answers torch.int32 cuda [P,B,N].  Accumulates into ``counters`` (int64 [counters_size(B)],
        allocated zeroed if None) and writes ``cells`` (uint8 [P,B,16], allocated if None; pass False
        to skip).  ``overwrite=True``: the counters are overwritten instead (no zeroing by the caller; with
        few long cells the whole evaluation is then ONE kernel launch).  Returns (counters, cells, cell_tokens).
        Does not synchronise."""
        P, B, N, counters, cells, cell_tokens, ptr, cptrs = self._device_call(answers, "answers", (0, 0), truth, tokens, n_valid, counters, cells,
                                                                              cell_tokens, overwrite)
        check(self._L.scv_aggregate_i32(self._ctx, ptr(answers), ptr(tokens), ptr(n_valid), ptr(truth), P, B, N, _lib.MEM_DEVICE,
                                        ptr(cells), ptr(cell_tokens), *cptrs))
        return counters, cells, cell_tokens

    def aggregate_bootstrap_device(self, answers, truth, r_begin: int, r_end: int, seed: int, M: int, tokens=None,
                                   n_valid=None, counters=None, cells=None, cell_tokens=None, out=None, overwrite=False):
        """Vote + problem-level bootstrap in ONE call (scv_aggregate_bootstrap_i32): ``aggregate_device`` followed by
        ``bootstrap_device`` over the cell table it writes -- in a single kernel launch when the shape allows it
        (all workgroups meet at a grid barrier after their last cell and share the resamples).  Returns
        (counters, cells, cell_tokens, boot int64 [r_end - r_begin, B, M]).  Asynchronous."""
        import torch
        P, B, N, counters, cells, cell_tokens, ptr, cptrs = self._device_call(answers, "answers", (0, 0), truth, tokens, n_valid, counters, cells,
                                                                              cell_tokens, overwrite, want_no_cells=False)
        if out is None:
            out = torch.empty((r_end - r_begin, B, M), dtype=torch.int64, device=answers.device)
        check(self._L.scv_aggregate_bootstrap_i32(self._ctx, ptr(answers), ptr(tokens), ptr(n_valid), ptr(truth), P, B, N, ptr(cells),
                                                  ptr(cell_tokens), *cptrs, r_begin, r_end, seed, M, ptr(out)))
        return counters, cells, cell_tokens, out

    def aggregate_prefix_device(self, pool, truth, n_valid, tokens=None, counters=None, cells=None, cell_tokens=None,
                                overwrite=False, budgets_host=None):
        """pool torch.int32 cuda [P,N], n_valid torch.int32 cuda [B].  Asynchronous; see aggregate_device.

        ``budgets_host``: the same budgets as a host sequence, when the caller has them (it usually built ``n_valid`` from one).
        The library cannot read device memory at launch time, so in auto mode a DEVICE-mode call over pools of 17 .. 128 votes queues
        two kernels that decide from ``n_valid`` which of them works (2-4 us); with the budgets known here and of the served form
        (``budgets_come_out_of_one_sort``) the call PROMISES them instead (option "prefix_path" = 5 for this call: one launch)."""
        P, B, N, counters, cells, cell_tokens, ptr, cptrs = self._device_call(pool, "pool", (0,), truth, tokens, n_valid, counters, cells,
                                                                              cell_tokens, overwrite)
        promise = False
        if budgets_host is not None and self._options.get("prefix_path", 0) == 0:
            if len(budgets_host) != B:
                raise ValueError("budgets_host must list the B budgets of n_valid")
            promise = self.budgets_come_out_of_one_sort(budgets_host, N)
        if promise:
            check(self._L.scv_set_option(self._ctx, b"prefix_path", 5))
        try:
            check(self._L.scv_aggregate_prefix_i32(self._ctx, ptr(pool), ptr(tokens), ptr(n_valid), ptr(truth), P, B, N, _lib.MEM_DEVICE,
                                                   ptr(cells), ptr(cell_tokens), *cptrs))
        finally:
            if promise:
                check(self._L.scv_set_option(self._ctx, b"prefix_path", 0))
        return counters, cells, cell_tokens

    def synth_fill_device(self, answers=None, tokens=None, truth=None, *, P, B, N, seed, dist, p_offset=0):
        """Fill preallocated CUDA int32 tensors with the closed-form synthetic data (asynchronous)."""
        self.use_torch_stream()
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        check(self._L.scv_synth_fill_i32(self._ctx, ptr(answers), ptr(tokens), ptr(truth), P, B, N, p_offset, seed, dist))

    def export_error_word(self, dst):
        """dst: int64 cuda tensor (>= 1 element, on the engine's device).  dst[0] = the device error word, written in
        stream order behind everything queued so far; not cleared (``sync`` does that).  No host round trip: the word
        can ride behind the packed counters in the evaluation's one all-reduce (scv_export_error_word)."""
        import torch
        if not (dst.is_cuda and dst.dtype == torch.int64 and dst.numel() >= 1):
            raise ValueError("dst must be an int64 CUDA tensor")
        self._check_device(dst, "dst")
        self.use_torch_stream()
        check(self._L.scv_export_error_word(self._ctx, C.c_void_p(dst.data_ptr())))

    def bootstrap_device(self, cells, r_begin: int, r_end: int, seed: int, M: int, out=None):
        """cells uint8 cuda [P,B,16] -> int64 cuda [r_end-r_begin, B, M] (asynchronous)."""
        import torch
        P, B = cells.shape[0], cells.shape[1]
        self._check_device(cells, "cells")
        self.use_torch_stream()
        if out is None:
            out = torch.empty((r_end - r_begin, B, M), dtype=torch.int64, device=cells.device)
        check(self._L.scv_bootstrap(self._ctx, C.c_void_p(cells.data_ptr()), P, B, r_begin, r_end, seed, M,
                                    _lib.MEM_DEVICE, C.c_void_p(out.data_ptr())))
        return out


class MultiDeviceEngine:
    """Single-process use of several GPUs, for a reference-style script (o1.py is one process): problems
    are sharded by contiguous block (dist.shard_bounds), each block goes through its own ``Engine`` on
    its own device from its own thread (the ctypes calls release the GIL; every GPU has its own PCIe
    link, so HOST-mode ingest scales with the device count), and the integer counters are summed on the
    host -- the same algebra as the RCCL all-reduce of the one-process-per-GPU path, bit-exact.

    ``devices``: list of HIP device indices (default: all visible).  The same index may appear twice
    (two contexts on one GPU), which is how the 1-GPU test box exercises this class.

    The contexts belong to ONE communicator of the library (``scv_comm_create``): DEVICE-mode evaluations finish with
    ``scv_allreduce_counters`` -- the exchange step behind the C ABI, no torch.distributed and no torch collective:
    a one-shot all-reduce over xGMI peer access (default), or RCCL single-process (``rccl=True``; distinct devices).
    """

    def __init__(self, devices=None, rccl: bool = False, timing: bool = False, clamp_to_invalid_bin: bool = False):
        L = _lib.load()
        if devices is None:
            n = L.scv_device_count()
            if n <= 0:
                raise _lib.ScvError(_lib.ERR_NO_DEVICE, "no HIP device visible")
            devices = list(range(n))
        devices = [int(d) for d in devices]
        self._L = L
        self._comm = C.c_void_p()
        flags = (_lib.FLAG_TIMING if timing else 0) | (_lib.FLAG_CLAMP if clamp_to_invalid_bin else 0)
        arr = (C.c_int * len(devices))(*devices)
        self.fell_back_to_rccl = False
        try:
            check(L.scv_comm_create(C.byref(self._comm), arr, len(devices), flags, _lib.COMM_RCCL if rccl else _lib.COMM_PEER))
        except _lib.ScvError as e:
            # First contact with a node's xGMI fabric: when the create-time self-test of the one-shot peer all-reduce fails (wrong words
            # read from a peer -- the message names the device pair), the reference-shaped one-process caller (o1.py:312-315) still gets
            # its result: the same communicator over single-process RCCL, with a warning that says why.
            if rccl or "self-test" not in str(e):
                raise
            import warnings
            warnings.warn(f"SCV_COMM_PEER is not usable on devices {devices} ({e}); falling back to SCV_COMM_RCCL", RuntimeWarning, stacklevel=2)
            self._comm = C.c_void_p()
            check(L.scv_comm_create(C.byref(self._comm), arr, len(devices), flags, _lib.COMM_RCCL))
            rccl = True
            self.fell_back_to_rccl = True
        self.rccl = bool(rccl)
        self.engines = [Engine(device=d, timing=timing, clamp_to_invalid_bin=clamp_to_invalid_bin, _adopt=L.scv_comm_ctx(self._comm, r))
                        for r, d in enumerate(devices)]

    def close(self):
        for e in getattr(self, "engines", []):
            e.close()                           # (adopted contexts: only forgets the pointer)
        if getattr(self, "_comm", None) is not None and self._comm:
            self._L.scv_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def all_reduce_counters(self, counters):
        """``counters[g]``: int64 CUDA tensor on engine g's device (equal lengths).  In-place SUM over the engines through the
        library's communicator (scv_allreduce_counters), ordered behind the engines' queued work; asynchronous."""
        n = int(counters[0].numel())
        for e, c in zip(self.engines, counters):
            if int(c.numel()) != n or not c.is_contiguous():
                raise ValueError("counters must be contiguous int64 tensors of one length")
            e._check_device(c, "counters")
        ptrs = (C.c_void_p * len(counters))(*[c.data_ptr() for c in counters])
        check(self._L.scv_allreduce_counters(self._comm, ptrs, n))
        return counters

    def stat(self, key: str) -> int:
        """Communicator counters (scv_comm_get_stat): "selftest_words" (verified per rank by the create-time self-test), "staging_bytes",
        "peer_loads" (0 nontemporal / 1 ordinary loads / -1 no peer reads), "selftest_nt_ok", "selftest_plain_ok"."""
        v = C.c_int64()
        check(self._L.scv_comm_get_stat(self._comm, key.encode(), C.byref(v)))
        return int(v.value)

    def all_gather_cells(self, tables, rows):
        """``tables[g]``: uint8 CUDA tensor [P, B, 16] on engine g's device -- the WHOLE cell table, in which engine g has written
        its own block of ``rows[g]`` problems at row ``sum(rows[:g])``.  In place: afterwards every table holds every block
        (scv_allgather_cells; asynchronous, ordered behind the engines' queued work)."""
        P, B = int(tables[0].shape[0]), int(tables[0].shape[1])
        if sum(int(r) for r in rows) != P or len(rows) != len(self.engines):
            raise ValueError("rows must hold one block size per engine and add up to the table's problems")
        for e, t in zip(self.engines, tables):
            if tuple(t.shape) != (P, B, 16) or not t.is_contiguous():
                raise ValueError("tables must be contiguous uint8 [P, B, 16] tensors of one shape")
            e._check_device(t, "tables")
        ptrs = (C.c_void_p * len(tables))(*[t.data_ptr() for t in tables])
        check(self._L.scv_allgather_cells(self._comm, ptrs, (C.c_int64 * len(rows))(*[int(r) for r in rows]), B))
        return tables

    def all_gather_i64(self, buffers, counts):
        """In-place all-gather of int64 blocks: engine g's block of ``counts[g]`` words sits at word ``sum(counts[:g])`` of
        ``buffers[g]`` (scv_allgather_i64)."""
        tot = sum(int(c) for c in counts)
        for e, b in zip(self.engines, buffers):
            if int(b.numel()) < tot or not b.is_contiguous():
                raise ValueError("buffers must be contiguous int64 tensors of at least sum(counts) words")
            e._check_device(b, "buffers")
        ptrs = (C.c_void_p * len(buffers))(*[b.data_ptr() for b in buffers])
        check(self._L.scv_allgather_i64(self._comm, ptrs, (C.c_int64 * len(counts))(*[int(c) for c in counts])))
        return buffers

    def evaluate_c5(self, shards, resamples: int, seed: int, M: int | None = None, n_valid=None, keep_all_ranks: bool = False):
        """BASELINE config 5 from ONE process over this communicator's GPUs, no torch.distributed (the
        multi-process form is passk.evaluate_device): ``shards[g] = (answers_g, truth_g, tokens_g | None)`` resident on engine g's
        device (``scatter`` makes them).  Per engine: vote over its block (cells written into its block of the whole table) ->
        ONE all-reduce of the packed counters + error word -> all-gather of the 16-byte cells -> bootstrap of its slice of the
        resamples -> all-gather of the slices.  ``M=None`` reads the class bound from the counters (one host sync).
        Returns (counters int64 [counters_size(B) + 1] -- the last word is the summed device error word --, cells uint8 [P, B, 16],
        boot int64 [resamples, B, M], M), all on the first engine's device and complete on EVERY engine; asynchronous: ``sync()``
        before reading (it raises what any engine has to report).  ``keep_all_ranks=True`` also keeps EVERY engine's copy of the
        three exchanged buffers in ``self.last_c5`` until the next call (bench.py verifies the exchange step on every rank with
        it); by default nothing outlives the call but what is returned (the copies held 3 x G buffers of HBM alive)."""
        import torch
        self.last_c5 = None
        if len(shards) != len(self.engines):
            raise ValueError("one shard per engine")
        G = len(self.engines)
        rows = [int(sh[0].shape[0]) for sh in shards]
        P, B = sum(rows), int(shards[0][0].shape[1])
        ncount = counters_size(B)
        counters, tables = [], []
        lo = 0
        for e, (ans, tr, tok), n in zip(self.engines, shards, rows):
            dev = ans.device
            with torch.cuda.device(dev):
                cnt = torch.zeros(ncount + 1, dtype=torch.int64, device=dev)
                table = torch.empty((P, B, 16), dtype=torch.uint8, device=dev)
                nv = None if n_valid is None else torch.as_tensor(np.asarray(n_valid, dtype=np.int32), device=dev)
                if n:
                    e.aggregate_device(ans, tr, tokens=tok, n_valid=nv, counters=cnt[:ncount], cells=table[lo:lo + n])
                e.export_error_word(cnt[ncount:])
            counters.append(cnt)
            tables.append(table)
            lo += n
        self.all_reduce_counters(counters)
        self.all_gather_cells(tables, rows)
        if M is None:
            from .passk import class_bound
            M = class_bound(counters[0][:ncount], B)             # host sync (the classes present do not depend on the resample seed)
        bounds = [((g * resamples) // G, ((g + 1) * resamples) // G) for g in range(G)]
        boots = []
        for e, table, (r0, r1) in zip(self.engines, tables, bounds):
            with torch.cuda.device(table.device):
                full = torch.empty((resamples, B, M), dtype=torch.int64, device=table.device)
                if r1 > r0:
                    e.bootstrap_device(table, r0, r1, seed, M, out=full[r0:r1])
            boots.append(full)
        self.all_gather_i64(boots, [(r1 - r0) * B * M for r0, r1 in bounds])
        # every engine's copy of the three exchanged buffers (each must be complete): what a caller that wants to verify the
        # exchange step on EVERY rank reads back (bench.py does, after its timed region)
        if keep_all_ranks:
            self.last_c5 = {"counters": counters, "tables": tables, "boots": boots}
        return counters[0], tables[0], boots[0], M

    def _run(self, fn_name, rows, truth, per_shard_kwargs, shared_kwargs):
        from concurrent.futures import ThreadPoolExecutor
        from .dist import shard_bounds
        P, G = rows.shape[0], len(self.engines)
        bounds = [shard_bounds(P, g, G) for g in range(G)]

        def work(g):
            lo, hi = bounds[g]
            if hi == lo:
                return None
            kw = {k: (None if v is None else v[lo:hi]) for k, v in per_shard_kwargs.items()}
            return getattr(self.engines[g], fn_name)(rows[lo:hi], truth[lo:hi], **kw, **shared_kwargs)

        with ThreadPoolExecutor(max_workers=G) as pool:
            parts = [r for r in pool.map(work, range(G)) if r is not None]
        if not parts:
            return getattr(self.engines[0], fn_name)(rows, truth, **per_shard_kwargs, **shared_kwargs)
        cat = lambda xs: None if any(x is None for x in xs) else np.concatenate(xs, axis=0)  # noqa: E731
        return AggregateResult(
            P, parts[0].B, cat([r.cells for r in parts]), cat([r.cell_tokens for r in parts]),
            sum(r.tie_class_hits for r in parts), sum(r.token_sum for r in parts),
            sum(r.truth_count_sum for r in parts))

    def aggregate(self, answers, truth, tokens=None, n_valid=None, want_cells=True) -> AggregateResult:
        answers = np.ascontiguousarray(answers, dtype=np.int32)
        truth = np.ascontiguousarray(truth, dtype=np.int32)
        tokens = None if tokens is None else np.ascontiguousarray(tokens, dtype=np.int32)
        return self._run("aggregate", answers, truth, {"tokens": tokens}, {"n_valid": n_valid, "want_cells": want_cells})

    # ---- DEVICE mode: shards already resident on their GPUs ------------------------------------------------

    def scatter(self, answers, truth, tokens=None):
        """Host arrays -> one shard per engine (contiguous blocks of problems, dist.shard_bounds), each on its
        engine's device: [(answers_g, truth_g, tokens_g)].  Convenience for callers that start from host data and
        then evaluate repeatedly on the device."""
        import torch
        from .dist import shard_bounds
        P, G = int(answers.shape[0]), len(self.engines)
        out = []
        for g, e in enumerate(self.engines):
            lo, hi = shard_bounds(P, g, G)
            dev = torch.device("cuda", e.device)
            cut = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x[lo:hi], dtype=np.int32)).to(dev)  # noqa: E731
            out.append((cut(answers), cut(truth), cut(tokens)))
        return out

    def aggregate_device(self, shards, n_valid=None, want_cells=True, destination=None):
        """Single-process, several GPUs, data resident: ``shards[g] = (answers_g, truth_g, tokens_g | None)`` are torch
        tensors on engine g's device (any split of the problems; ``scatter`` makes block shards).  Every engine launches
        on its own device's current stream (asynchronously, from this one thread), then the packed int64 counters
        (65.7 KB at B = 8) are all-reduced in place by the library's communicator (``scv_allreduce_counters``: one-shot over
        xGMI peer access, or RCCL) -- every engine's buffer ends up holding the sum; the one on ``destination`` (default:
        the first engine's device) is returned.  Integer sums: bit-exact, order-independent.
        Returns (counters, [cells_g], [cell_tokens_g]); does not synchronise the host.
        ``n_valid``: host int array [B] (copied to every device) or None."""
        import torch
        if len(shards) != len(self.engines):
            raise ValueError("one shard per engine")
        outs = []
        for e, (ans, tr, tok) in zip(self.engines, shards):
            dev = ans.device
            with torch.cuda.device(dev):
                nv = None if n_valid is None else torch.as_tensor(np.asarray(n_valid, dtype=np.int32), device=dev)
                outs.append(e.aggregate_device(ans, tr, tokens=tok, n_valid=nv, cells=None if want_cells else False))
        counters = [o[0] for o in outs]
        self.all_reduce_counters(counters)          # every engine's buffer now holds the sum (scv_allreduce_counters)
        total = counters[0]
        if destination is not None:
            dest = torch.device(destination)
            match = [c for c in counters if c.device == dest]
            total = match[0] if match else counters[0].to(dest)
        return total, [o[1] for o in outs], [o[2] for o in outs]

    def sync(self):
        """scv_comm_sync: every engine's stream drained, every engine's device error word judged (first error raised)."""
        check(self._L.scv_comm_sync(self._comm))

    def aggregate_prefix(self, pool, truth, n_valid, tokens=None, want_cells=True) -> AggregateResult:
        pool = np.ascontiguousarray(pool, dtype=np.int32)
        truth = np.ascontiguousarray(truth, dtype=np.int32)
        tokens = None if tokens is None else np.ascontiguousarray(tokens, dtype=np.int32)
        return self._run("aggregate_prefix", pool, truth, {"tokens": tokens}, {"n_valid": n_valid, "want_cells": want_cells})


def unpack_cells(packed_u32: np.ndarray) -> np.ndarray:
    """SCV_FLAG_PACKED_CELLS records (include/scvote.h: max_count | truth_count << 7 | n_modes << 14 | min_mode << 21 | hit << 31) -> CELL_DTYPE,
    field for field what the 16-byte record of the same cell holds (min_mode = -1 for an empty cell)."""
    w = np.ascontiguousarray(packed_u32).astype(np.uint32, copy=False)
    out = np.zeros(w.shape, dtype=CELL_DTYPE)
    out["max_count"] = w & 0x7F
    out["truth_count"] = (w >> 7) & 0x7F
    out["n_modes"] = (w >> 14) & 0x7F
    mm = ((w >> 21) & 0x3FF).astype(np.int16)
    out["min_mode"] = np.where(out["max_count"] == 0, np.int16(-1), mm)
    out["hit"] = (w >> 31).astype(np.uint8)
    return out


def cells_from_torch(cells_u8) -> np.ndarray:
    """uint8 cuda/cpu [P,B,16] (or [P,B,4]: packed records of an Engine(packed_cells=True)) -> CELL_DTYPE [P,B] on the host."""
    arr = np.ascontiguousarray(cells_u8.detach().cpu().numpy())
    if arr.shape[-1] == 4:
        return unpack_cells(arr.view(np.uint32).reshape(arr.shape[0], arr.shape[1]))
    return arr.view(CELL_DTYPE).reshape(arr.shape[0], arr.shape[1])
