// scvote_dispatch.h -- host-side tables that map a launch geometry to a kernel instantiation.
//
// The kernel family is ~120 template instantiations.  They are spread over several translation units
// (scvote_stream_c{4,8,16}.hip: the streaming kernel by LDS replication; scvote_reg_g{8,16,32,64}.hip: register-resident
// cells by lanes per cell; scvote_dense.hip: register-streamed long cells; scvote_sort.hip: sorted cells) so that hipcc compiles
// them in parallel (_build.py) and an edit to one kernel rebuilds one table.  scvote.hip (the C ABI) only sees these prototypes.
#pragma once

#include "scvote_kernels.hip.h"

namespace scv {

using KernelFn = void (*)(const AggArgs);
struct RegKernel { KernelFn fn; int waves; };   // + the workgroup size (waves) the kernel was compiled for

// Waves per workgroup of a one-wave-per-step kernel (scv_sort_cells, scv_sort_prefix, scv_sort_prefix2) when the launch has fewer steps than the chip
// has wave slots: every SIMD of a CU gets a wave before any gets a second one (two waves of a SIMD in the same VALU-bound step take turns: 4.9 against
// 3.0 us for the sort of a one-step launch of 64-vote pools) -- but never fewer than one wave per SIMD: every workgroup ends in one device atomic per
// counter word, ~13 ns each on ONE word whoever sends it (391 one-wave workgroups: 5 us of epilogue; 98 of four waves: 1.3; profiles/r06_sort_prefix_wall.log).
// Only while that at least halves the workgroup: 7 waves instead of 8 leave three SIMDs with two waves AND add workgroups (1e5 pools of 64 votes: 22.5 against 21.7 us).
// (least: 4 = one wave per SIMD; 8 for the cheap steps of scv_sort_cells<8 | 16>, where more workgroups cost more than shared SIMDs: 1000 steps of 16 votes 9.6 against 8.9 us)
inline int spread_waves(int64_t nsteps, int waves, int num_cus, int least = 4) {
    int w = (int)((nsteps + num_cus - 1) / num_cus);
    if (w < least) w = least;
    return 2 * w <= waves ? w : waves;
}

// streaming kernel scv_hist_argmax<log2(copies), threads, unroll, tokens, xtra>.  Instantiated geometries (copies, threads):
// (4, 256) (8, 256) (8, 512) (16, 256) (16, 512) (16, 1024), 4 loads in flight per lane ((4, 256): 2 -- the short-cell band);
// xtra (single-launch epilogues: overwrite-counters, bootstrap behind a grid barrier) for the four the library picks itself:
// (4, 256) (8, 256) (16, 512) (16, 1024).  NULL: not instantiated.
KernelFn pick_kernel(int copies, int threads, int unroll, bool tok, bool xtra);
KernelFn pick_stream_c4(int threads, int unroll, bool tok, bool xtra);
KernelFn pick_stream_c8(int threads, int unroll, bool tok, bool xtra);
KernelFn pick_stream_c16(int threads, int unroll, bool tok, bool xtra);

// scv_reg_cells<g lanes per cell, v vectors per lane, ...>: capacity 4 * g * v votes per cell; the shapes the dispatch uses:
// (16, 1) (8, 3) (16, 2) (16, 4) (32, 4) (64, 4)
RegKernel pick_reg_kernel(int g, int v, bool tok, bool vec);
RegKernel pick_reg_g8(int v, bool tok, bool vec);      // (8, 3): 65 ... 96 votes in 96 slots, 8-bit bins; (8, 4): an A/B shape for 97 ... 128
RegKernel pick_reg_g16(int v, bool tok, bool vec);
RegKernel pick_reg_g32(int v, bool tok, bool vec);
RegKernel pick_reg_g64(int v, bool tok, bool vec);

// scv_reg_dense<v vectors per lane per part, h parts>: capacity 256 * v * h votes per cell
RegKernel pick_dense_kernel(int v, int h, bool tok, bool vec);

// scv_sort_cells<nv votes per lane>: one lane per cell, 4 <= N <= nv, rows staged by LDS-DMA (scvote_sort.hip.h);
// .waves = the launch bound in waves
RegKernel pick_sort_kernel(int nv, bool tok, bool lin);

// scv_prefix_pool<g lanes per problem, v vectors per lane>: prefix budgets over one pool row per problem, every budget out of one pass
// (scvote_prefix.hip.h); shapes (16, 4) and (32, 4); .waves = the launch bound in waves; words of LDS per wave = prefix_pool_hist_words(g) +
// kPrefixPoolLaneWords; behind the waves ord[B] | nvs[B] | kPrefixPoolFixedWords (the head map) | the counter tables
constexpr int kPrefixPoolLaneWords = 64;          // behind a wave's histograms: one trash word per lane
constexpr int kPrefixPoolFixedWords = 32;
#ifndef SCV_PREFIX_H16
#define SCV_PREFIX_H16 1
#endif
constexpr bool prefix_pool_h16(int g) { return SCV_PREFIX_H16 != 0 && g == 16; }           // 16-bit bins, two per word
constexpr int prefix_pool_hist_words(int g) { return (64 / g) * (prefix_pool_h16(g) ? kBins / 2 : kBins); }
RegKernel pick_prefix_pool_kernel(int g, bool tok, bool vec);

// scv_sort_prefix<nv votes per lane>: prefix budgets that are powers of two (and the whole row) over pools of nv / 2 < N <= nv votes, out of
// one sort per problem (scvote_sort_prefix.hip.h); .waves = the launch bound in waves; LDS words behind the waves' regions:
// sort_prefix_tail_words(nv, B)
constexpr int sort_prefix_classes(int nv) { int l = 0; while ((1 << l) < nv / 2) ++l; return l + 3; }
constexpr long long sort_prefix_tail_words(int nv, int B) { return 16 + ((B + 3) & ~3) + ((sort_prefix_classes(nv) * (nv + 1) + 1) & ~1) + 4 * sort_prefix_classes(nv); }
RegKernel pick_sort_prefix_kernel(int nv, bool tok);

// ---- shared by the table translation units ------------------------------------------------------------------------
template <int RL2, int T, int U>
inline KernelFn stream_tok(bool tok, bool xtra) {
    if (xtra) return tok ? (KernelFn)scv_hist_argmax<RL2, T, U, true, true> : (KernelFn)scv_hist_argmax<RL2, T, U, false, true>;
    return tok ? (KernelFn)scv_hist_argmax<RL2, T, U, true> : (KernelFn)scv_hist_argmax<RL2, T, U, false>;
}
template <int RL2, int T, int U>
inline KernelFn stream_plain(bool tok) {
    return tok ? (KernelFn)scv_hist_argmax<RL2, T, U, true> : (KernelFn)scv_hist_argmax<RL2, T, U, false>;
}
template <int G, int V, int K>
inline RegKernel reg_gv(bool tok, bool vec) {
    if (tok) return vec ? RegKernel{(KernelFn)scv_reg_cells<G, V, K, true, true>, reg_cells_waves<G, V, true, true>()}
                        : RegKernel{(KernelFn)scv_reg_cells<G, V, K, true, false>, reg_cells_waves<G, V, true, false>()};
    return vec ? RegKernel{(KernelFn)scv_reg_cells<G, V, K, false, true>, reg_cells_waves<G, V, false, true>()}
               : RegKernel{(KernelFn)scv_reg_cells<G, V, K, false, false>, reg_cells_waves<G, V, false, false>()};
}

}  // namespace scv
