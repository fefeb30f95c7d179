// Register-resident cells, 32 lanes per cell (scv_reg_cells<32, V, K, ...>, V = 1, 2, 4 vectors per lane).
#include "scvote_dispatch.h"
namespace scv {
RegKernel pick_reg_g32(int v, bool tok, bool vec) { return reg_g<32>(v, tok, vec); }
}  // namespace scv
