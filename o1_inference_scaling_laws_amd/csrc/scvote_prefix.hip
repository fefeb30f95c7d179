// Prefix budgets over one sample pool per problem, every budget out of one pass (scv_prefix_pool<G lanes per problem, V vectors per lane, TOK, VEC>).
#include "scvote_prefix.hip.h"
#include "scvote_dispatch.h"
namespace scv {
// rows that are not 16-byte aligned are read with dword loads, each with an address of its own: chunks of 2 vectors per lane instead of 4
template <int G>
static RegKernel pool_g(bool tok, bool vec) {
    if (tok) return vec ? RegKernel{(KernelFn)scv_prefix_pool<G, 4, true, true>, prefix_pool_waves<G, true>()} : RegKernel{(KernelFn)scv_prefix_pool<G, 2, true, false>, prefix_pool_waves<G, true>()};
    return vec ? RegKernel{(KernelFn)scv_prefix_pool<G, 4, false, true>, prefix_pool_waves<G, false>()} : RegKernel{(KernelFn)scv_prefix_pool<G, 2, false, false>, prefix_pool_waves<G, false>()};
}
// g: lanes per problem: 16 (a chunk of 4 vectors per lane = 256 votes: such a row is held whole) or 32 (512); vec: every pool row 16-byte aligned
RegKernel pick_prefix_pool_kernel(int g, bool tok, bool vec) {
    if (g == 16) return pool_g<16>(tok, vec);
    return pool_g<32>(tok, vec);
}
}  // namespace scv
