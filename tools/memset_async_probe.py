#!/usr/bin/env python3
"""Is hipMemset on DEVICE memory asynchronous to the host?  (Round 6: the split-N scratch used to be cleared by hipMemset -- the null stream --
right before a launch on the context's own NON-BLOCKING stream; profiles/r06_split_scratch_race.log.)  Times the call's return against the
device's completion for fills of 256 MiB ... 4 GiB: a call that returns in far less than bytes / HBM rate has not waited.  ctypes only."""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
def chk(rc):
    assert rc == 0, rc
chk(hip.hipSetDevice(0))
for mib in (256, 1024, 4096):
    n = mib << 20
    p = C.c_void_p()
    chk(hip.hipMalloc(C.byref(p), C.c_size_t(n)))
    chk(hip.hipMemset(p, 0, C.c_size_t(n)))          # first touch (page mapping) outside the measurement
    chk(hip.hipDeviceSynchronize())
    best_call, best_total = 1e9, 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        chk(hip.hipMemset(p, 1, C.c_size_t(n)))
        t1 = time.perf_counter()
        chk(hip.hipDeviceSynchronize())
        t2 = time.perf_counter()
        best_call, best_total = min(best_call, t1 - t0), min(best_total, t2 - t0)
    print(f"hipMemset {mib:5d} MiB: call returns after {best_call * 1e6:9.1f} us, device done after {best_total * 1e6:9.1f} us "
          f"(= {n / best_total / 1e9:6.0f} GB/s) -> {'ASYNCHRONOUS to the host' if best_call < 0.5 * best_total else 'waits for completion'}")
    chk(hip.hipFree(p))
