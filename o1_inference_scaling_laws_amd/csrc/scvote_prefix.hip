// Prefix budgets over one sample pool per problem, every budget out of one pass (scv_prefix_pool<G lanes per problem, V vectors per lane, TOK, VEC>).
#include "scvote_prefix.hip.h"
#include "scvote_dispatch.h"
namespace scv {
// rows that are not 16-byte aligned are read with dword loads, each with an address of its own: chunks of 2 vectors per lane instead of 4, and
// always 16 lanes per problem (the host asks for no other unaligned shape: the 32-lane unaligned form gave its two kernel slots to
// scv_sort_prefix2 -- unaligned pools of more than 1024 votes run 100 instead of 87 us at 2e4 x 4096)
template <int G>
static RegKernel pool_vec(bool tok) {
    return tok ? RegKernel{(KernelFn)scv_prefix_pool<G, 4, true, true>, prefix_pool_waves<G, true>()} : RegKernel{(KernelFn)scv_prefix_pool<G, 4, false, true>, prefix_pool_waves<G, false>()};
}
// g: lanes per problem: 16 (a chunk of 4 vectors per lane = 256 votes: such a row is held whole) or 32 (512); vec: every pool row 16-byte aligned
RegKernel pick_prefix_pool_kernel(int g, bool tok, bool vec) {
    if (!vec) return tok ? RegKernel{(KernelFn)scv_prefix_pool<16, 2, true, false>, prefix_pool_waves<16, true>()} : RegKernel{(KernelFn)scv_prefix_pool<16, 2, false, false>, prefix_pool_waves<16, false>()};
    if (g == 16) return pool_vec<16>(tok);
    return pool_vec<32>(tok);
}
}  // namespace scv
