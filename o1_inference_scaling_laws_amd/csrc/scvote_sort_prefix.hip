// Prefix budgets that are powers of two over short pools, out of ONE sort per problem (scv_sort_prefix<NV, TOK>).
#include "scvote_sort_prefix.hip.h"
#include "scvote_dispatch.h"
namespace scv {
// nv: votes per lane (32 / 64; 128: the two-phase kernel scv_sort_prefix2, with tokens: + its token steps); .waves = the launch bound in waves
RegKernel pick_sort_prefix_kernel(int nv, bool tok) {
    if (nv == 128) return tok ? RegKernel{(KernelFn)scv_sort_prefix2<true>, sort_prefix2_threads() / 64} : RegKernel{(KernelFn)scv_sort_prefix2<false>, sort_prefix2_threads() / 64};
    if (nv == 32) return tok ? RegKernel{(KernelFn)scv_sort_prefix<32, true>, sort_prefix_threads(32) / 64} : RegKernel{(KernelFn)scv_sort_prefix<32, false>, sort_prefix_threads(32) / 64};
    return tok ? RegKernel{(KernelFn)scv_sort_prefix<64, true>, sort_prefix_threads(64) / 64} : RegKernel{(KernelFn)scv_sort_prefix<64, false>, sort_prefix_threads(64) / 64};
}
}  // namespace scv
#ifdef SCV_SP_TIMELINE
// measurement build only: read (and clear) the phase sums of scv_sort_prefix
extern "C" int scv_debug_sort_prefix_timeline(unsigned long long* out8, int clear) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(scv::scv_sp_timeline), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (clear) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(scv::scv_sp_timeline), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef SCV_SP_WALL
// the wall-clock marks of the last launch (scv_sp_wall: a row of 16 per wave), reduced: [0] = min of the starts, [i] = max over the waves that passed mark i; clear = all 0
extern "C" int scv_debug_sort_prefix_wall(unsigned long long* out16, int clear) {
    static unsigned long long h[scv::kSpWallWaves * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(scv::scv_sp_wall), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) out16[i] = i == 0 ? ~0ull : 0ull;
    for (int w = 0; w < scv::kSpWallWaves; ++w)
        for (int i = 0; i < 16; ++i) {
            const unsigned long long v = h[w * 16 + i];
            if (!v) continue;
            if (i == 0) out16[0] = v < out16[0] ? v : out16[0];
            else out16[i] = v > out16[i] ? v : out16[i];
        }
    // [11 ..15]: the SIMD of waves 0 .. 7 of workgroups 0 and 1 (HW_ID bits 5:4), one hex digit per wave; [12] = of workgroup 100, 101
    for (int g = 0; g < 2; ++g) {
        unsigned long long m = 0;
        for (int w = 0; w < 16; ++w) m |= ((h[((g ? 800 : 0) + w) * 16 + 10] >> 4) & 3ull) << (4 * w);
        out16[11 + g] = m;
    }
    if (clear) {
        for (auto& v : h) v = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(scv::scv_sp_wall), h, sizeof h) != hipSuccess) return -1;
    }
    return 0;
}
#endif
