// Register-resident cells, 16 lanes per cell (scv_reg_cells<16, V, K, ...>, V = 1, 2, 4 vectors per lane).
#include "scvote_dispatch.h"
namespace scv {
RegKernel pick_reg_g16(int v, bool tok, bool vec) { return reg_g<16>(v, tok, vec); }
}  // namespace scv
