// Streaming kernel scv_hist_argmax with R = 16 LDS copies of the histogram (the headline geometry: 1024 threads, one workgroup per
// CU): 256 / 512 / 1024 threads, 4 loads in flight per lane; the single-launch epilogues for 512 and 1024 threads.
#include "scvote_dispatch.h"
namespace scv {
KernelFn pick_stream_c16(int threads, int unroll, bool tok, bool xtra) {
    if (unroll != 4) return nullptr;
    if (threads == 1024) return stream_tok<4, 1024, 4>(tok, xtra);
    if (threads == 512) return stream_tok<4, 512, 4>(tok, xtra);
    if (threads == 256 && !xtra) return stream_plain<4, 256, 4>(tok);
    return nullptr;
}
}  // namespace scv
