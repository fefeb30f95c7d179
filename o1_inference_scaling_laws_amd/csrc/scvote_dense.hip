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
    return h == 1 ? dense_vh<4, 1>(tok, vec) : (h == 2 ? dense_vh<4, 2>(tok, vec) : dense_vh<4, 4>(tok, vec));
}
}  // namespace scv
