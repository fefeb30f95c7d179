// Register-streamed long cells: V vectors per lane per part, H parts per cell (capacity 256 * V * H votes), dense bin scan.
#include "scvote_dispatch.h"
namespace scv {
template <int V, int H>
static RegKernel dense_vh(bool tok, bool vec) {
    if (tok) return vec ? RegKernel{(KernelFn)scv_reg_dense<V, H, true, true>, reg_dense_waves<V, H, true, true>()}
                        : RegKernel{(KernelFn)scv_reg_dense<V, H, true, false>, reg_dense_waves<V, H, true, false>()};
    return vec ? RegKernel{(KernelFn)scv_reg_dense<V, H, false, true>, reg_dense_waves<V, H, false, true>()}
               : RegKernel{(KernelFn)scv_reg_dense<V, H, false, false>, reg_dense_waves<V, H, false, false>()};
}
// (8-vector parts -- 8 KiB in flight per wave -- were measured slower than 4-vector parts in round 2 (200 VGPRs, 79 vs 72 us at
//  N = 2048) and are no longer instantiated: with the pivot counters the token variants no longer fit the register file.)
RegKernel pick_dense_kernel(int v, int h, bool tok, bool vec) {
    (void)v;
    switch (h) {
    case 1: return dense_vh<4, 1>(tok, vec);
    case 2: return dense_vh<4, 2>(tok, vec);
    case 8: return dense_vh<4, 8>(tok, vec);       // 4096 < N <= 8192: the streaming kernel's 1024 * R-word fold per cell still costs more than 8 parts
                                                   // through the wave's own 8 KiB histogram (N = 4608: 4.5 -> 5.4 TB/s; from 8192 up they meet, 16 parts lose;
                                                   // and a count of 2^14 would not fit the key = count << 18 | address)
    default: return dense_vh<4, 4>(tok, vec);
    }
}
}  // namespace scv
