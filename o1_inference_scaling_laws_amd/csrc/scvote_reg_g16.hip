// Register-resident cells, 16 lanes per cell (scv_reg_cells<16, V, K, ...>, V = 1, 2, 4 vectors per lane; short cells run K = 4 / V
// batches per iteration: 4 KiB of votes in flight per wave behind the batch being counted -- 8 and 16 KiB were measured equal / slower).
#include "scvote_dispatch.h"
namespace scv {
RegKernel pick_reg_g16(int v, bool tok, bool vec) {
    return v == 1 ? reg_gv<16, 1, 4>(tok, vec) : (v == 2 ? reg_gv<16, 2, 2>(tok, vec) : reg_gv<16, 4, 1>(tok, vec));
}
}  // namespace scv
