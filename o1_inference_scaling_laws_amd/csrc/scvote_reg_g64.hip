// Register-resident cells, one wave per cell (scv_reg_cells<64, V, K, ...>) + the A/B dense scan with 32-bit bins.
#include "scvote_dispatch.h"
namespace scv {
RegKernel pick_reg_g64(int v, bool tok, bool vec, bool dense4) {
    if (v == 4 && dense4) return reg_gv<64, 4, 1, true>(tok, vec);
    return reg_g<64>(v, tok, vec);
}
}  // namespace scv
