// Short cells, one lane per cell, rows staged through LDS by LDS-DMA and sorted in registers (scv_sort_cells<NV, TOK, LIN>).
#include "scvote_sort.hip.h"
#include "scvote_dispatch.h"
namespace scv {
template <int NV, bool LIN>
static RegKernel sort_n(bool tok) {
    return tok ? RegKernel{(KernelFn)scv_sort_cells<NV, true, LIN>, sort_cells_threads(NV) / 64}
               : RegKernel{(KernelFn)scv_sort_cells<NV, false, LIN>, sort_cells_threads(NV) / 64};
}
// nv: votes per lane (8 / 16 / 24 / 32 / 40 / 48 / 56 / 64; 24, 40, 56: round 6 -- 17 ... 24 votes used to sort 32 slots, 33 ... 40 votes 48, 49 ... 56 votes 64); lin: rows that are not all 16-byte aligned (linear image, dword reads)
RegKernel pick_sort_kernel(int nv, bool tok, bool lin) {
    if (lin) {
        switch (nv) {
        case 8: return sort_n<8, true>(tok);
        case 16: return sort_n<16, true>(tok);
        case 24: return sort_n<24, true>(tok);
        case 32: return sort_n<32, true>(tok);
        case 40: return sort_n<40, true>(tok);
        case 48: return sort_n<48, true>(tok);
        case 56: return sort_n<56, true>(tok);
        default: return sort_n<64, true>(tok);
        }
    }
    switch (nv) {
    case 8: return sort_n<8, false>(tok);
    case 16: return sort_n<16, false>(tok);
    case 24: return sort_n<24, false>(tok);
    case 32: return sort_n<32, false>(tok);
    case 40: return sort_n<40, false>(tok);
    case 48: return sort_n<48, false>(tok);
    case 56: return sort_n<56, false>(tok);
    default: return sort_n<64, false>(tok);
    }
}
}  // namespace scv
#ifdef SCV_SORT_TIMELINE
// measurement build only: read (and clear) the phase sums of scv_sort_cells
extern "C" int scv_debug_sort_timeline(unsigned long long* out8, int clear) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(scv::scv_sort_timeline), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (clear) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(scv::scv_sort_timeline), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

