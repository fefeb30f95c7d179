// Register-resident cells, one wave per cell (scv_reg_cells<64, 4, 1, ...>: 513 ... 896 votes).
#include "scvote_dispatch.h"
namespace scv {
RegKernel pick_reg_g64(int v, bool tok, bool vec) { (void)v; return reg_gv<64, 4, 1>(tok, vec); }
}  // namespace scv
