// scvote_dispatch.h -- host-side tables that map a launch geometry to a kernel instantiation.
//
// The kernel family is ~200 template instantiations.  They are spread over several translation units
// (scvote_stream_c{4,8,16,32}.hip: the streaming kernel by LDS replication; scvote_reg_g{16,32,64}.hip: register-resident
// cells by lanes per cell; scvote_dense.hip: register-streamed long cells) so that hipcc compiles them in parallel
// (_build.py) and an edit to one kernel rebuilds one table.  scvote.hip (the C ABI) only sees these prototypes.
#pragma once

#include "scvote_kernels.hip.h"

namespace scv {

using KernelFn = void (*)(const AggArgs);
struct RegKernel { KernelFn fn; int waves; };   // + the workgroup size (waves) the kernel was compiled for

// streaming kernel scv_hist_argmax<log2(copies), threads, unroll, tokens, xtra>; xtra: single-launch epilogues (unroll 4 only)
KernelFn pick_kernel(int copies, int threads, int unroll, bool tok, bool xtra);
KernelFn pick_stream_c4(int threads, int unroll, bool tok, bool xtra);
KernelFn pick_stream_c8(int threads, int unroll, bool tok, bool xtra);
KernelFn pick_stream_c16(int threads, int unroll, bool tok, bool xtra);
KernelFn pick_stream_c32(int threads, int unroll, bool tok, bool xtra);

// scv_reg_cells<g lanes per cell, v vectors per lane, ...>: capacity 4 * g * v votes per cell
RegKernel pick_reg_kernel(int g, int v, bool tok, bool vec, bool dense4, int km);
RegKernel pick_reg_g16(int v, bool tok, bool vec);
RegKernel pick_reg_g32(int v, bool tok, bool vec);
RegKernel pick_reg_g64(int v, bool tok, bool vec, bool dense4);

// scv_reg_dense<v vectors per lane per part, h parts>: capacity 256 * v * h votes per cell
RegKernel pick_dense_kernel(int v, int h, bool tok, bool vec);

// scv_sort_cells<nv votes per lane, kb blocks of 64 cells per step>: one lane per cell, 4 <= N <= nv, rows staged by LDS-DMA
// (scvote_sort.hip.h); .waves = the launch bound in waves
RegKernel pick_sort_kernel(int nv, int kb, bool tok, bool lin, bool db);

// ---- shared by the table translation units ------------------------------------------------------------------------
template <int RL2, int T, int U>
inline KernelFn stream_tok(bool tok) {
    return tok ? (KernelFn)scv_hist_argmax<RL2, T, U, true> : (KernelFn)scv_hist_argmax<RL2, T, U, false>;
}
template <int RL2, int T>
inline KernelFn stream_u(int u, bool tok) {
    switch (u) {
    case 2: return stream_tok<RL2, T, 2>(tok);
    case 8: return stream_tok<RL2, T, 8>(tok);
    default: return stream_tok<RL2, T, 4>(tok);
    }
}
template <int RL2, int T>
inline KernelFn stream_xtra(bool tok) {
    return tok ? (KernelFn)scv_hist_argmax<RL2, T, 4, true, true> : (KernelFn)scv_hist_argmax<RL2, T, 4, false, true>;
}
template <int RL2>
inline KernelFn stream_t(int t, int u, bool tok, bool xtra) {
    switch (t) {
    case 256: return xtra ? stream_xtra<RL2, 256>(tok) : stream_u<RL2, 256>(u, tok);
    case 1024: return xtra ? stream_xtra<RL2, 1024>(tok) : stream_u<RL2, 1024>(u, tok);
    default: return xtra ? stream_xtra<RL2, 512>(tok) : stream_u<RL2, 512>(u, tok);
    }
}
template <int G, int V, int K, bool DENSE>
inline RegKernel reg_gv(bool tok, bool vec) {
    if (tok) return vec ? RegKernel{(KernelFn)scv_reg_cells<G, V, K, true, true, DENSE>, reg_cells_waves<G, V, true, DENSE, true>()}
                        : RegKernel{(KernelFn)scv_reg_cells<G, V, K, true, false, DENSE>, reg_cells_waves<G, V, true, DENSE, false>()};
    return vec ? RegKernel{(KernelFn)scv_reg_cells<G, V, K, false, true, DENSE>, reg_cells_waves<G, V, false, DENSE, true>()}
               : RegKernel{(KernelFn)scv_reg_cells<G, V, K, false, false, DENSE>, reg_cells_waves<G, V, false, DENSE, false>()};
}
// lanes per cell g, 16-byte vectors per lane v (capacity 4*g*v votes); short cells run K = 4 / v batches per
// iteration (4 KiB of votes in flight per wave behind the batch being counted -- measured: 8 KiB is slower, the
// waves wait on LDS, not on memory; the 8 and 16 KiB variants are not instantiated).
template <int G>
inline RegKernel reg_g(int v, bool tok, bool vec) {
    return v == 1 ? reg_gv<G, 1, 4, false>(tok, vec) : (v == 2 ? reg_gv<G, 2, 2, false>(tok, vec) : reg_gv<G, 4, 1, false>(tok, vec));
}

}  // namespace scv
