// Short cells, one lane per cell, rows staged through LDS by LDS-DMA and sorted in registers (scv_sort_cells<NV, KB, TOK>).
#include "scvote_sort.hip.h"
#include "scvote_dispatch.h"
namespace scv {
template <int NV, int KB>
static RegKernel sort_nk(bool tok) {
    return tok ? RegKernel{(KernelFn)scv_sort_cells<NV, KB, true>, sort_cells_threads(NV) / 64}
               : RegKernel{(KernelFn)scv_sort_cells<NV, KB, false>, sort_cells_threads(NV) / 64};
}
// nv: votes per lane (8 / 16 / 32 / 64); kb: blocks of 64 cells per step (2 only for nv <= 16)
RegKernel pick_sort_kernel(int nv, int kb, bool tok) {
    switch (nv) {
    case 8: return kb == 2 ? sort_nk<8, 2>(tok) : sort_nk<8, 1>(tok);
    case 16: return kb == 2 ? sort_nk<16, 2>(tok) : sort_nk<16, 1>(tok);
    case 32: return sort_nk<32, 1>(tok);
    default: return sort_nk<64, 1>(tok);
    }
}
}  // namespace scv
