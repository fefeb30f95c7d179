// Prefix budgets that are powers of two over short pools, out of ONE sort per problem (scv_sort_prefix<NV, TOK>).
#include "scvote_sort_prefix.hip.h"
#include "scvote_dispatch.h"
namespace scv {
// nv: votes per lane (32 / 64; 128: the two-phase kernel scv_sort_prefix2); .waves = the launch bound in waves
RegKernel pick_sort_prefix_kernel(int nv, bool tok) {
    if (nv == 128) return RegKernel{(KernelFn)scv_sort_prefix2, sort_prefix2_threads() / 64};                  // (no tokens: see launch_prefix)
    if (nv == 32) return tok ? RegKernel{(KernelFn)scv_sort_prefix<32, true>, sort_prefix_threads(32) / 64} : RegKernel{(KernelFn)scv_sort_prefix<32, false>, sort_prefix_threads(32) / 64};
    return tok ? RegKernel{(KernelFn)scv_sort_prefix<64, true>, sort_prefix_threads(64) / 64} : RegKernel{(KernelFn)scv_sort_prefix<64, false>, sort_prefix_threads(64) / 64};
}
// scv_prefix_tokens<lanes per row>: the token sums of prefix budgets over pools of up to 64 / 128 tokens per row
static_assert(kPrefixTokensU == kPrefixTokensGroups, "the host sizes the token kernel's LDS from kPrefixTokensGroups");
KernelFn pick_prefix_tokens_kernel(int lanes) { (void)lanes; return (KernelFn)scv_prefix_tokens<32>; }   // (the 16-lane form -- rows of up to 64 tokens -- is not instantiated: those pools send their tokens through scv_sort_prefix's image)
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
