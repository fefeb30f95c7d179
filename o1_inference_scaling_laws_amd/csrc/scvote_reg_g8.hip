// Register-resident cells, 8 lanes per cell, 8-bit bins (scv_reg_cells<8, V, K, ...>; round 6): 65 ... 96 votes occupy the 96 slots of
// 8 lanes x 3 vectors and 97 ... 128 votes 8 lanes x 4 vectors instead of the 128 slots of 16 lanes x 2 -- eight cells per wave in the same
// 8 KiB of LDS.  K: batches per loop iteration ((8, 3): two batches = 6 KiB of votes in flight per wave behind the pair being counted).
// Same-box A/B against round 5's library (N = 72 / 96, 2e5 x 4 cells): 106.2 -> 82.7 / 110.3 -> 83.2 us; (8, 4) against (16, 2): N = 100 / 112 /
// 125 / 128: 112.2 -> 101.9 / 112.8 -> 104.9 / 118.4 -> 107.9 / 105.6 -> 102.2 us.
#include "scvote_dispatch.h"
#ifndef SCV_G8_K
#define SCV_G8_K 2
#endif
namespace scv {
RegKernel pick_reg_g8(int v, bool tok, bool vec) {
    if (v == 3) return reg_gv<8, 3, SCV_G8_K>(tok, vec);
    if (v == 4) return reg_gv<8, 4, 1>(tok, vec);
    return RegKernel{nullptr, 0};
}
}  // namespace scv
