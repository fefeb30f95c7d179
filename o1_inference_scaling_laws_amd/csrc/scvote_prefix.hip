// Prefix budgets over one sample pool per problem, every budget out of one pass (scv_prefix_pool<G lanes per problem, TOK, VEC>).
#include "scvote_prefix.hip.h"
#include "scvote_dispatch.h"
namespace scv {
template <int G>
static RegKernel pool_g(bool tok, bool vec) {
    if (tok) return vec ? RegKernel{(KernelFn)scv_prefix_pool<G, true, true>, prefix_pool_waves<G>()} : RegKernel{(KernelFn)scv_prefix_pool<G, true, false>, prefix_pool_waves<G>()};
    return vec ? RegKernel{(KernelFn)scv_prefix_pool<G, false, true>, prefix_pool_waves<G>()} : RegKernel{(KernelFn)scv_prefix_pool<G, false, false>, prefix_pool_waves<G>()};
}
// g: lanes per problem (16 / 32 / 64); vec: every pool row 16-byte aligned
RegKernel pick_prefix_pool_kernel(int g, bool tok, bool vec) {
    if (g == 16) return pool_g<16>(tok, vec);
    if (g == 32) return pool_g<32>(tok, vec);
    return pool_g<64>(tok, vec);
}
}  // namespace scv
