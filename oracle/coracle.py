"""TEST INFRASTRUCTURE -- ctypes binding of oracle/scv_oracle.c (the C restatement of
/root/reference/o1.py:181-247).  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libscv_oracle.so")

CELL_DTYPE = np.dtype(
    [("max_count", "<u4"), ("truth_count", "<u4"), ("n_modes", "<u2"), ("min_mode", "<i2"),
     ("hit", "u1"), ("pad", "u1", (3,))]
)
assert CELL_DTYPE.itemsize == 16
TIE_CLASSES = 1025


def build(force: bool = False) -> str:
    """Compile scv_oracle.c -> libscv_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "scv_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libscv_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        p = C.c_void_p
        L.scvo_aggregate_i32.argtypes = [p, p, p, p, C.c_int64, C.c_int32, C.c_int64, C.c_int, p, p, p, p, p]
        L.scvo_aggregate_i32.restype = C.c_int
        L.scvo_aggregate_i32_mt.argtypes = [p, p, p, p, C.c_int64, C.c_int32, C.c_int64, C.c_int, C.c_int, p, p, p, p, p]
        L.scvo_aggregate_i32_mt.restype = C.c_int
        L.scvo_synth_fill_i32.argtypes = [p, p, p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_uint64, C.c_int]
        L.scvo_synth_fill_i32.restype = C.c_int
        L.scvo_bootstrap.argtypes = [p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, p]
        L.scvo_bootstrap.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def aggregate(answers, truth, tokens=None, n_valid=None, clamp=False):
    """answers int32[P,B,N]; returns dict of the integer outputs of include/scvote.h."""
    answers = np.ascontiguousarray(answers, dtype=np.int32)
    P, B, N = answers.shape
    truth = np.ascontiguousarray(truth, dtype=np.int32)
    assert truth.shape == (P,)
    if tokens is not None:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        assert tokens.shape == answers.shape
    if n_valid is not None:
        n_valid = np.ascontiguousarray(n_valid, dtype=np.int32)
        assert n_valid.shape == (B,)
    cells = np.zeros((P, B), dtype=CELL_DTYPE)
    cell_tokens = np.zeros((P, B), dtype=np.int64)
    tie = np.zeros((B, TIE_CLASSES), dtype=np.int64)
    tok = np.zeros((B,), dtype=np.int64)
    tcs = np.zeros((B,), dtype=np.int64)
    rc = lib().scvo_aggregate_i32(_ptr(answers), _ptr(tokens), _ptr(n_valid), _ptr(truth), P, B, N,
                                  int(bool(clamp)), _ptr(cells), _ptr(cell_tokens), _ptr(tie), _ptr(tok), _ptr(tcs))
    return {"rc": rc, "cells": cells, "cell_tokens": cell_tokens, "tie_class_hits": tie,
            "token_sum": tok, "truth_count_sum": tcs}


def aggregate_mt(answers, truth, threads, tokens=None, n_valid=None, clamp=False):
    """aggregate() with the problems spread over `threads` OpenMP threads (bench.py all-cores baseline)."""
    answers = np.ascontiguousarray(answers, dtype=np.int32)
    P, B, N = answers.shape
    truth = np.ascontiguousarray(truth, dtype=np.int32)
    if tokens is not None:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
    if n_valid is not None:
        n_valid = np.ascontiguousarray(n_valid, dtype=np.int32)
    cells = np.zeros((P, B), dtype=CELL_DTYPE)
    cell_tokens = np.zeros((P, B), dtype=np.int64)
    tie = np.zeros((B, TIE_CLASSES), dtype=np.int64)
    tok = np.zeros((B,), dtype=np.int64)
    tcs = np.zeros((B,), dtype=np.int64)
    rc = lib().scvo_aggregate_i32_mt(_ptr(answers), _ptr(tokens), _ptr(n_valid), _ptr(truth), P, B, N, int(bool(clamp)),
                                     int(threads), _ptr(cells), _ptr(cell_tokens), _ptr(tie), _ptr(tok), _ptr(tcs))
    return {"rc": rc, "cells": cells, "cell_tokens": cell_tokens, "tie_class_hits": tie,
            "token_sum": tok, "truth_count_sum": tcs}


def synth_fill(P, B, N, seed, dist, p_offset=0, want_tokens=False):
    answers = np.empty((P, B, N), dtype=np.int32)
    tokens = np.empty((P, B, N), dtype=np.int32) if want_tokens else None
    truth = np.empty((P,), dtype=np.int32)
    rc = lib().scvo_synth_fill_i32(_ptr(answers), _ptr(tokens), _ptr(truth), P, B, N, p_offset, seed, dist)
    assert rc == 0, rc
    return answers, tokens, truth


def synth_truth(P, seed, p_offset=0):
    truth = np.empty((P,), dtype=np.int32)
    rc = lib().scvo_synth_fill_i32(None, None, _ptr(truth), P, 1, 1, p_offset, seed, 0)
    assert rc == 0, rc
    return truth


def bootstrap(cells, r_begin, r_end, seed, M):
    cells = np.ascontiguousarray(cells)
    assert cells.dtype == CELL_DTYPE and cells.ndim == 2
    P, B = cells.shape
    out = np.zeros((r_end - r_begin, B, M), dtype=np.int64)
    rc = lib().scvo_bootstrap(_ptr(cells), P, B, r_begin, r_end, seed, M, _ptr(out))
    return rc, out
