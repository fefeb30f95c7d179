// Short cells, one lane per cell, rows staged through LDS by LDS-DMA and sorted in registers (scv_sort_cells<NV, KB, TOK, LIN>).
#include "scvote_sort.hip.h"
#include "scvote_dispatch.h"
namespace scv {
template <int NV, int KB, bool LIN, bool DB = false>
static RegKernel sort_nk(bool tok) {
    if constexpr (NV == 128) return RegKernel{(KernelFn)scv_sort_cells<NV, KB, false, LIN, DB>, sort_cells_threads(NV) / 64};   // (votes only)
    else return tok ? RegKernel{(KernelFn)scv_sort_cells<NV, KB, true, LIN, DB>, sort_cells_threads(NV) / 64}
                    : RegKernel{(KernelFn)scv_sort_cells<NV, KB, false, LIN, DB>, sort_cells_threads(NV) / 64};
}
// nv: votes per lane (8 / 16 / 32 / 64 / 128); kb: blocks of 64 cells per step (2 only for aligned rows and nv <= 16);
// lin: rows that are not all 16-byte aligned (linear image, dword reads); db: two buffers per wave (nv <= 16, kb = 1)
RegKernel pick_sort_kernel(int nv, int kb, bool tok, bool lin, bool db) {
    if (db && nv <= 16) {                                             // two image buffers per wave, the copy two steps ahead (short rows)
        if (lin) return nv == 8 ? sort_nk<8, 1, true, true>(tok) : sort_nk<16, 1, true, true>(tok);
        return nv == 8 ? sort_nk<8, 1, false, true>(tok) : sort_nk<16, 1, false, true>(tok);
    }
    if (lin) {
        switch (nv) {
        case 8: return sort_nk<8, 1, true>(tok);
        case 16: return sort_nk<16, 1, true>(tok);
        case 32: return sort_nk<32, 1, true>(tok);
        default: return sort_nk<64, 1, true>(tok);
        }
    }
    switch (nv) {
    case 8: return kb == 2 ? sort_nk<8, 2, false>(tok) : sort_nk<8, 1, false>(tok);
    case 16: return kb == 2 ? sort_nk<16, 2, false>(tok) : sort_nk<16, 1, false>(tok);
    case 32: return sort_nk<32, 1, false>(tok);
    case 128: return sort_nk<128, 1, false>(tok);
    default: return sort_nk<64, 1, false>(tok);
    }
}
}  // namespace scv
