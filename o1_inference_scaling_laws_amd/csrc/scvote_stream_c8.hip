// Streaming kernel scv_hist_argmax with R = 8 LDS copies of the histogram: 3 workgroup sizes x (3 unrolls + the
// single-launch-epilogue variant) x tokens.  One table per translation unit (scvote_dispatch.h).
#include "scvote_dispatch.h"
namespace scv {
KernelFn pick_stream_c8(int threads, int unroll, bool tok, bool xtra) { return stream_t<3>(threads, unroll, tok, xtra); }
}  // namespace scv
