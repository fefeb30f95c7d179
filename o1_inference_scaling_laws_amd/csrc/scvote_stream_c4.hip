// Streaming kernel scv_hist_argmax with R = 4 LDS copies of the histogram (short cells, many workgroups per CU): 256 threads,
// 2 loads in flight per lane -- the geometry the library picks below 4096 votes per cell when the register-resident kernels are off.
#include "scvote_dispatch.h"
namespace scv {
KernelFn pick_stream_c4(int threads, int unroll, bool tok, bool xtra) {
    if (threads != 256 || unroll != 2) return nullptr;
    return stream_tok<2, 256, 2>(tok, xtra);
}
}  // namespace scv
