// Register-resident cells, 32 lanes per cell (scv_reg_cells<32, 4, 1, ...>: 257 ... 512 votes).
#include "scvote_dispatch.h"
namespace scv {
RegKernel pick_reg_g32(int v, bool tok, bool vec) { (void)v; return reg_gv<32, 4, 1>(tok, vec); }
}  // namespace scv
