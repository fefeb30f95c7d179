// Streaming kernel scv_hist_argmax with R = 8 LDS copies of the histogram: 256 threads (+ the single-launch epilogues), 512 threads.
#include "scvote_dispatch.h"
namespace scv {
KernelFn pick_stream_c8(int threads, int unroll, bool tok, bool xtra) {
    if (unroll != 4) return nullptr;
    if (threads == 256) return stream_tok<3, 256, 4>(tok, xtra);
    if (threads == 512 && !xtra) return stream_plain<3, 512, 4>(tok);
    return nullptr;
}
}  // namespace scv
