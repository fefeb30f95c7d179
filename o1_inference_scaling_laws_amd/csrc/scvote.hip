// scvote.hip -- C ABI (include/scvote.h) over the gfx950 kernels in scvote_kernels.hip.h.
//
// Host-side responsibilities: argument validation, launch geometry (persistent grid sized from
// the CU count), variant dispatch for A/B tuning, hipEvent timing on the launch stream, and the
// HOST-memory staging path.  No C++ exception crosses the ABI; every entry returns 0 or a
// negative code and sets a thread-local message.
#define SCV_TU_MAIN 1   // the non-template kernels are emitted by this translation unit only
#include "scvote_kernels.hip.h"
#include "scvote_dispatch.h"
#include "scvote_hostpool.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// No C++ exception crosses the C ABI (include/scvote.h): every extern "C" body runs inside guarded().  What can throw
// on the host side is the standard library (std::vector / std::function growth: std::bad_alloc; std::thread creation in a container
// with a thread limit: std::system_error); all of it becomes an error code + message, never std::terminate in a ctypes caller.
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(SCV_ERR_ALLOC, "out of host memory (std::bad_alloc inside the library)");
    } catch (const std::exception& e) {
        return fail(SCV_ERR_ARG, "internal error: %s", e.what());
    } catch (...) {
        return fail(SCV_ERR_ARG, "internal error: unknown C++ exception");
    }
}

// TEST HOOK, compiled only with -DSCV_TEST_HOOKS (the `hooks` variant of _build.py: csrc/libscvote_hooks.so, which only
// tests/test_gpu_parity.py loads; libscvote.so, the product, never reads the environment): SCV_TEST_FAULT makes the HOST-mode staging
// path fail the way a starved container would -- "thread": every worker-thread creation raises std::system_error (the pipeline must
// run on the calling thread alone); "alloc": std::bad_alloc while the copy pieces are built; "throw": a std::runtime_error.
int test_fault() {
#ifdef SCV_TEST_HOOKS
    const char* f = getenv("SCV_TEST_FAULT");
    if (!f || !*f) return 0;
    if (!strcmp(f, "thread")) return 1;
    if (!strcmp(f, "alloc")) return 2;
    if (!strcmp(f, "throw")) return 3;
    if (!strcmp(f, "race")) return 4;          // a DELIBERATE data race among the copy pieces: proves that a sanitizer run would see one
#endif
    return 0;
}

#define SCV_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? SCV_ERR_ALLOC : -(int)e_, "%s: %s", #expr,     \
                        hipGetErrorString(e_));                                                    \
    } while (0)

struct EventPair { hipEvent_t a, b; };

}  // namespace

// ---- HOST-mode ingestion pipeline -----------------------------------------------------
// Real (non-synthetic) answer tensors arrive in pageable host memory, and then the 63 GB/s PCIe link -- not HBM --
// bounds end-to-end votes/s.  A pageable hipMemcpyAsync goes through the runtime's own bounce buffers with one
// copying thread (measured 38-46 GB/s, and it blocks the host).  Here the chunk loop is a three-stage pipeline:
//   worker threads   memcpy chunk i+1 of the caller's buffers into a PINNED bounce slot   (several cores)
//   copy stream      DMA slot -> HBM slot of chunk i                                       (one 55+ GB/s transfer)
//   compute stream   hot-path kernel(s) on chunk i-1, cell table back to the caller
// with two bounce slots, two HBM slots and events between the stages.  Buffers the caller has already pinned
// (hipHostMalloc / hipHostRegister: scv_host_alloc) skip the bounce and are DMA'd in place.
struct HostPipe {
    static constexpr int kSlots = 2;
    hipStream_t copy_stream = nullptr;
    hipEvent_t landed[kSlots] = {nullptr, nullptr};      // H2D of the slot finished (bounce slot reusable, kernel may start)
    hipEvent_t consumed[kSlots] = {nullptr, nullptr};    // kernel + cell D2H on the slot finished (HBM slot reusable)
    void* bounce[kSlots] = {nullptr, nullptr};           // pinned host memory
    size_t bounce_bytes = 0;
    void* dslot[kSlots] = {nullptr, nullptr};            // HBM
    size_t dslot_bytes = 0;
    // worker threads for the pageable -> pinned copies: plain C++ with no HIP in it (scvote_hostpool.h), so that gcc's
    // -fsanitize=thread / address can be pointed at it on the CPU (tests/hostpool_sanitize.cpp)
    scv::CopyPool pool;
    void start(int nthreads, bool fail_for_test = false) noexcept { pool.start(nthreads, fail_for_test); }
    void run(std::vector<std::function<void()>>&& pieces) { pool.run(std::move(pieces)); }
    void shutdown() {
        pool.join_all();
        for (int k = 0; k < kSlots; ++k) {
            if (landed[k]) (void)hipEventDestroy(landed[k]);
            if (consumed[k]) (void)hipEventDestroy(consumed[k]);
            if (bounce[k]) (void)hipHostFree(bounce[k]);
            if (dslot[k]) (void)hipFree(dslot[k]);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

struct scv_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    int64_t lds_max = 65536;
    int64_t clock_khz = 0;
    int64_t hbm_bytes = 0;
    // streaming-kernel geometry (scv_set_tuning; auto unless user_tuned)
    int copies = 16, threads = 1024, wg_per_cu = 1, unroll = 4;
    bool user_tuned = false; // set_tuning called: auto geometry off
    // options (scv_set_option; include/scvote.h documents every key)
    int path = 0;            // 0 auto | 1 streaming, whole cells | 2 streaming, split-N | 4 register-resident cells | 5 sorted cells
    int segs_override = 0;   // > 0: segments per cell for path 2
    int split_seg_kb = 256;  // split-N: smallest segment the auto choice cuts (KiB)
    int overwrite_counters = 0;  // DEVICE mode: per-budget outputs are overwritten instead of accumulated into (no caller memset)
    int sort_n_min = 8;      // sorted cells (scv_sort_cells): sort_n_min <= N <= sort_n_max (rows that are not 16-byte aligned: from 5); shorter
    int sort_n_max = 64;     // cells stay on scv_lane_cells, longer ones go to the register-resident kernels; sort_n_max = 0: off
    int reg_n_max = 8192;    // auto: 32 < N <= this -> register-resident cells (scv_reg_cells / scv_reg_dense); 0 = off (streaming kernel)
    int reg_shape = 0;       // force a register-resident shape (parity tests, A/B runs), see launch_aggregate
    int fused_counters_max = 512;   // cells: at or below, counters inside the hot kernel; above, scv_reduce_cells; 0: always the reduction.  (Round 6, tools/crossovers.py:
                                    // 4096 until then -- at 2048 / 4096 / 8192 cells of 64 KiB the per-cell atomics cost 41 / 73 / 138 us against 29 / 50 / 89 with the reduction;
                                    // equal at 512 cells, 4 % better at 256)
    int grid_override = 0;   // > 0: exact persistent grid size
    int prefix_path = 0;     // prefix budgets: 0 auto | 1 one lane per problem | 2 cell kernels on pool rows | 3 one-pass streaming snapshots |
                             // 4 one pass per problem, G lanes per problem (scv_prefix_pool; "reg_shape" 16 / 32 / 64 forces G)
    int boot_path = 0;       // vote + bootstrap: 0 auto (one cooperative launch when the shape allows) | 1 one ORDINARY launch | 2 two launches,
                             // LDS-resident code table | 3 two launches, global gathers
    int boot_spin_limit = 1 << 20;   // polls (x s_sleep 8) a workgroup waits at the grid barrier before giving up (tests force 1)
    int stage_mb = 128;      // HOST mode: chunk size (votes + tokens) of the staging pipeline
    int copy_threads = 6;    // HOST mode (4-8 reach the link rate; 16+ were unstable: 30-55 GB/s run to run): threads copying pageable caller memory
    // fixed choices that used to be options (measured: DESIGN_HISTORY.md 4 "dead ends")
    static constexpr int tiny_n_max = 32;   // N <= this (and below sort_n_min): one lane per cell, registers only (scv_lane_cells)
    void* d_tickets = nullptr;   // arrival counters of the single-launch modes (all zero between launches)
    size_t d_tickets_words = 0;
    int64_t stat_boot_fused = 0, stat_boot_separate = 0, stat_overwrite_fused = 0, stat_lds_counters = 0, stat_prefix_cells = 0, stat_prefix_lane = 0;   // scv_get_stat
    struct BootReq { int32_t r0, r1, M; uint64_t seed; int64_t* out; bool fused; }* boot_req = nullptr;   // set for the duration of one call
    // the last fused request, kept so that a grid-barrier timeout can be repaired at scv_sync by a separate bootstrap launch
    struct BootLast { const scv_cell* cells = nullptr; int64_t P = 0; int32_t B = 0, r0 = 0, r1 = 0, M = 0; uint64_t seed = 0; int64_t* out = nullptr; bool valid = false; } boot_last;
    int64_t stat_boot_recovered = 0, stat_boot_cooperative = 0, stat_sort_cells = 0, stat_few_votes = 0, stat_prefix_pool = 0;
    int64_t stat_prefix_sort = 0, stat_prefix_tokens = 0, stat_one_vote = 0;
    const int32_t* nv_host = nullptr;   // HOST-mode calls: the caller's n_valid (host memory) for the duration of the call -- launch_prefix reads the budgets
    // split-N scratch (grown on demand)
    void* d_partial = nullptr;
    size_t d_partial_bytes = 0;
    void* d_cells = nullptr;  // cell table scratch for the reduce kernel when the caller wants no cells
    size_t d_cells_bytes = 0;
    // device scratch
    uint32_t* d_err = nullptr;
    bool err_dirty = false;
    // timing
    std::vector<EventPair> events;
    size_t events_used = 0;
    // HOST-mode staging (grown on demand)
    void* d_stage = nullptr;
    size_t d_stage_bytes = 0;
    HostPipe* pipe = nullptr;       // HOST-mode ingestion pipeline (created on the first HOST call)
    // HOST-mode small calls (everything the reference itself asks for: P = 30, N <= 128, o1.py:277,302): one pinned block and one
    // HBM block owned by the ctx, allocated on the first small call
    void* small_h = nullptr;
    void* small_d = nullptr;
    size_t small_bytes = 0;
    int small_call_kb = 1024;       // option "host_small_kb": calls whose inputs + outputs fit in this many KiB take the small path (0: never)
    int64_t stat_small_calls = 0, stat_pipelined_calls = 0;
};

namespace {

constexpr size_t kMaxTimedLaunches = 8192;   // timing ring: undrained records beyond this are dropped

// Next hipEvent pair of the timing ring (NULL without SCV_FLAG_TIMING).
int next_event_pair(scv_ctx* ctx, EventPair** out) {
    *out = nullptr;
    if (!(ctx->flags & SCV_FLAG_TIMING)) return SCV_OK;
    if (ctx->events_used >= kMaxTimedLaunches) ctx->events_used = 0;
    if (ctx->events_used == ctx->events.size()) {
        EventPair np;
        SCV_HIP(hipEventCreate(&np.a));
        SCV_HIP(hipEventCreate(&np.b));
        ctx->events.push_back(np);
    }
    *out = &ctx->events[ctx->events_used++];
    return SCV_OK;
}


// Kernel variant tables live in their own translation units (scvote_stream_*.hip, scvote_reg.hip, scvote_dense.hip):
// the template instantiations are compiled in parallel and only the table that changed is rebuilt (csrc/scvote_dispatch.h).
using scv::KernelFn;
using scv::RegKernel;
using scv::pick_kernel;
using scv::pick_reg_kernel;
using scv::pick_dense_kernel;
using scv::pick_sort_kernel;
using scv::pick_prefix_pool_kernel;
using scv::pick_sort_prefix_kernel;

// Every entry point runs on the ctx device and leaves the caller's current HIP device as it found it
// (a process driving several GPUs from one thread -- MultiDeviceEngine -- must not have torch's
// current device moved under it).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev != device) {
            SCV_HIP(hipSetDevice(device));
            switched = true;
        }
        return SCV_OK;
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};
#define SCV_ENTER(ctx)                                                                             \
    DeviceGuard guard_;                                                                            \
    if (int rc_ = guard_.enter((ctx)->device)) return rc_

// Per-cell device atomics land on ~B addresses and serialise at the memory side.  Measured
// (profiles/r01_crossover_r4_d1.log): 65536 cells of 64 KiB with fused atomics run at 4.1 TB/s, with
// the separate reduction at 6.6 TB/s -- the serialised tail is NOT hidden behind the stream.  Keep the
// counters fused in the one launch only for few cells, or when a problem row is >= 4 MiB (the headline).
bool reduce_counters_separately(const scv_ctx* ctx, int64_t ncells, int32_t B, int64_t N) {
    if (ctx->fused_counters_max == 0) return true;                       // forced (tests)
    return ncells > ctx->fused_counters_max && N * (int64_t)B < (1 << 20);
}

int ensure_cells(scv_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->d_cells_bytes) return SCV_OK;
    if (ctx->d_cells) { SCV_HIP(hipFree(ctx->d_cells)); ctx->d_cells = nullptr; ctx->d_cells_bytes = 0; }
    SCV_HIP(hipMalloc(&ctx->d_cells, bytes));
    ctx->d_cells_bytes = bytes;
    return SCV_OK;
}

// Arrival counters of the single-launch modes (overwrite-counters, vote + bootstrap): 4 words, zero when allocated at scv_create,
// and every counter is reset by the workgroup that completes it, so they are all-zero again whenever no launch is in flight.
// Nothing allocates on the launch path: legal inside hipGraph capture and independent of later scv_set_stream calls.
constexpr size_t kTicketWords = 8 + scv::kSplitTickets;     // [0 .. 2] the single-launch epilogues | [8 + cell] split-N arrival counters

// split-N scratch: the split cells' histograms and token sums in memory.  All zero whenever no launch is in flight (the workgroup that
// finishes a cell clears what it read), so it is cleared here once, when it is (re)allocated -- ON THE CONTEXT'S STREAM: the stream is
// non-blocking, so a hipMemset (null stream, asynchronous to the host for device memory) is not ordered before the launch that follows and
// could clear sums the first segments had already added (round 6: fuzz seeds 116 / 155 -- calls in which the scratch grows -- failed in a test
// selection without a DEVICE-mode call in front; in the full suite such a call binds the context to torch's null stream early, which ordered the two:
// profiles/r06_split_scratch_race.log).
int ensure_partial(scv_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->d_partial_bytes) return SCV_OK;
    if (ctx->d_partial) { SCV_HIP(hipStreamSynchronize(ctx->stream)); SCV_HIP(hipFree(ctx->d_partial)); ctx->d_partial = nullptr; ctx->d_partial_bytes = 0; }
    SCV_HIP(hipMalloc(&ctx->d_partial, bytes));
    SCV_HIP(hipMemsetAsync(ctx->d_partial, 0, bytes, ctx->stream));
    ctx->d_partial_bytes = bytes;
    return SCV_OK;
}

// Launch the hot path on device pointers.  Accumulates into the per-budget counters.
//
// Regimes behind one entry point (auto-selected from the shape; the "path" option forces one):
//   lane      N <= 4 (and pool rows <= 32)  one lane per cell, registers only                (scv_lane_cells)
//   sorted    5 / 8 <= N <= 64              one lane per cell, rows by LDS-DMA, sorted       (scv_sort_cells)
//   register  up to 8192                    a cell in the registers of 16 / 32 / 64 lanes    (scv_reg_cells, scv_reg_dense)
//   split-N   cells <= CUs/2, big N         several workgroups per cell, merged in the launch  (scv_hist_argmax: device atomics + a ticket per cell)
//   stream    everything else               one persistent workgroup streams whole cells     (scv_hist_argmax)
// plus scv_reduce_cells behind the streaming kernel when the per-budget counters are not fused.
// lds bytes of the one-lane-per-cell kernel (tie classes 0..nv and two sums per budget)
size_t lane_kernel_lds(int32_t B, int nv) { return (((size_t)B * (nv + 1) + 1) & ~(size_t)1) * sizeof(uint32_t) + 2 * (size_t)B * sizeof(unsigned long long); }

// pool_rows: prefix budgets over one pool [P, N] through the CELL kernels (register-resident / one lane per cell):
// cell (p, b) counts the first n_valid[b] votes of row p.  Only launch_prefix passes it, after pool_rows_eligible().
int launch_aggregate(scv_ctx* ctx, const int32_t* answers, const int32_t* tokens, const int32_t* n_valid,
                     const int32_t* truth, int64_t P, int32_t B, int64_t N, scv_cell* cells,
                     int64_t* cell_tokens, int64_t* tie, int64_t* tok_sum, int64_t* truth_sum, bool pool_rows = false) {
    const int64_t ncells = P * (int64_t)B;
    if (ncells == 0) return SCV_OK;
    scv::AggArgs a;
    a.pool_rows = pool_rows ? 1 : 0;
    a.answers = answers; a.tokens = tokens; a.n_valid = n_valid; a.truth = truth;
    a.ncells = ncells; a.N = N; a.B = B; a.P = P;
    a.cells = cells; a.cell_tokens = cell_tokens;
    a.tie_hits = reinterpret_cast<unsigned long long*>(tie);
    a.token_sum = reinterpret_cast<unsigned long long*>(tok_sum);
    a.truth_sum = reinterpret_cast<unsigned long long*>(truth_sum);
    a.err_flag = ctx->d_err;
    a.prefetch = 1;
    a.sorted = 1;
    a.segs = 1; a.seg_len = N; a.partial = nullptr; a.partial_tok = nullptr; a.wave_lds_words = 0; a.acc_classes = 0;
    a.tickets = nullptr; a.overwrite = 0; a.ow_tie = a.ow_tok = a.ow_truth = nullptr; a.boot = 0; a.boot_r0 = a.boot_r1 = 0; a.boot_M = 1; a.boot_spins = 0; a.boot_seed = 0; a.boot_out = nullptr;
    a.packed_cells = ((ctx->flags & SCV_FLAG_PACKED_CELLS) && cells && !pool_rows) ? 1 : 0;   // (aggregate_common has checked the shape: N <= 127, cell kernels only)
    const bool tok = tokens != nullptr;
    const bool want_counters = tie || truth_sum || (tok && tok_sum);
    const bool rows_aligned = (N % 4 == 0) && (((uintptr_t)answers & 15u) == 0) && (!tok || ((uintptr_t)tokens & 15u) == 0);
    const int64_t Nreg = rows_aligned ? N : N + 3;     // register-resident kernels read unaligned rows as their aligned supersets (up to 3 slots more)

    // ---- which kernel (one decision, before anything is set up for it) -------------------------------------------------------
    enum Kind { LANE, SORT, REG, STREAM } kind;
    {
        const bool sort_ok = !pool_rows && ctx->sort_n_max > 0 && N >= (rows_aligned ? 4 : 1) && N <= (ctx->sort_n_max < 64 ? ctx->sort_n_max : 64);
        if (ctx->path == 5) kind = sort_ok ? SORT : (N <= ctx->tiny_n_max ? LANE : (Nreg <= 8192 ? REG : STREAM));
        else if (ctx->path == 4) kind = (Nreg <= 8192 && N >= 1) ? REG : STREAM;
        else if (ctx->path == 1 || ctx->path == 2) kind = STREAM;
        else if (sort_ok && N >= (rows_aligned ? ctx->sort_n_min : (ctx->sort_n_min < 5 ? ctx->sort_n_min : 5))) kind = SORT;
        else if (N <= ctx->tiny_n_max) kind = LANE;
        else if (N <= ctx->reg_n_max && Nreg <= 8192) kind = REG;
        else kind = STREAM;
        if (kind == LANE && lane_kernel_lds(B, N <= 4 ? 4 : (N <= 8 ? 8 : (N <= 16 ? 16 : 32))) > (size_t)60 * 1024) kind = REG;   // (thousands of budgets)
        // 17 .. 32 votes with tokens on one lane per cell need 256 VGPRs + scratch at 512 threads (pool rows / sorted cells switched off
        // only -- dense cells of this length are sorted): the register-resident shape of 64 votes takes them
        if (kind == LANE && tok && N > 16) kind = REG;
    }

    // ---- per-budget counters: inside the launch (LDS tables of the cell kernels; per-cell atomics of the streaming kernel for few
    // cells) or from the cell table by scv_reduce_cells (same-address device atomics serialise at ~12 ns each; measured,
    // profiles/r01_crossover_r4_d1.log: 65536 cells of 64 KiB with fused atomics 4.1 TB/s, with the separate reduction 6.6)
    bool use_reduce = want_counters && (ctx->fused_counters_max == 0 || (kind == STREAM && ncells > ctx->fused_counters_max && N * (int64_t)B < (1 << 20)));
    auto need_cell_scratch = [&]() -> int {          // the counters will be computed from the cell table: make sure there is one
        a.tie_hits = nullptr; a.token_sum = nullptr; a.truth_sum = nullptr;
        if (!a.cells || (tok && tok_sum && !a.cell_tokens)) {
            const size_t cb = (size_t)ncells * sizeof(scv_cell);
            if (int rc = ensure_cells(ctx, cb + (size_t)ncells * sizeof(int64_t) + 256)) return rc;
            if (!a.cells) a.cells = static_cast<scv_cell*>(ctx->d_cells);
            if (tok && !a.cell_tokens) a.cell_tokens = reinterpret_cast<int64_t*>(static_cast<char*>(ctx->d_cells) + ((cb + 255) / 256) * 256);
        }
        return SCV_OK;
    };
    // Overwrite semantics (option "overwrite_counters"): the streaming kernel turns its cell table into the counters with its
    // last workgroup (single launch: no memset, no reduce launch) when the cells are few; every other kernel gets a memset node
    // in front and accumulates as usual.
    bool overwrite_fused = false;
    if (ctx->overwrite_counters && want_counters) {
        if (kind == STREAM && ncells <= 8192 && B <= 64) {
            overwrite_fused = true;
            use_reduce = false;
        } else {
            if (tie) SCV_HIP(hipMemsetAsync(tie, 0, (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t), ctx->stream));
            if (tok_sum) SCV_HIP(hipMemsetAsync(tok_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
            if (truth_sum) SCV_HIP(hipMemsetAsync(truth_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
        }
    }
    if (use_reduce || overwrite_fused)
        if (int rc = need_cell_scratch()) return rc;
    if (overwrite_fused) {
        a.overwrite = 1;
        a.tickets = static_cast<uint32_t*>(ctx->d_tickets);
        a.ow_tie = reinterpret_cast<unsigned long long*>(tie);
        a.ow_tok = reinterpret_cast<unsigned long long*>(tok_sum);
        a.ow_truth = reinterpret_cast<unsigned long long*>(truth_sum);
    }
    auto finish = [&](EventPair* ev_) -> int {
        if (use_reduce) {
            int64_t chunks = (P + 2047) / 2048;
            const int64_t cap = ((int64_t)ctx->num_cus * 8 + B - 1) / B;
            if (chunks > cap) chunks = cap;
            if (chunks < 1) chunks = 1;
            auto* th = reinterpret_cast<unsigned long long*>(tie);
            auto* ts = reinterpret_cast<unsigned long long*>(tok_sum);
            auto* tc = reinterpret_cast<unsigned long long*>(truth_sum);
            // token reads only when token_sum is wanted: cell_tokens scratch exists exactly then
            if (tok && tok_sum) hipLaunchKernelGGL((scv::scv_reduce_cells<true>), dim3((unsigned)chunks, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, ctx->stream, a.cells, a.cell_tokens, P, B, th, ts, tc);
            else hipLaunchKernelGGL((scv::scv_reduce_cells<false>), dim3((unsigned)chunks, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, ctx->stream, a.cells, a.cell_tokens, P, B, th, ts, tc);
            SCV_HIP(hipGetLastError());
        }
        if (ev_) SCV_HIP(hipEventRecord(ev_->b, ctx->stream));
        ctx->err_dirty = true;
        return SCV_OK;
    };

    EventPair* ev = nullptr;
    if (int rc = next_event_pair(ctx, &ev)) return rc;

    if (kind == SORT) {
        // ---- sorted cells: one lane per cell, the wave's 64 rows staged through LDS by LDS-DMA, sorted in registers (scvote_sort.hip.h).
        // The reference's own range (N = 1 ... 128, o1.py:267,276).  Rows that are not all 16-byte aligned (N % 4 != 0, unaligned
        // bases) take the linear-image form (dword reads).
        const bool sort_lin = !rows_aligned;
        const int nv = N <= 8 ? 8 : (N <= 16 ? 16 : (N <= 24 ? 24 : (N <= 32 ? 32 : (N <= 40 ? 40 : (N <= 48 ? 48 : (N <= 56 ? 56 : 64))))));
        const RegKernel rk = pick_sort_kernel(nv, tok, sort_lin);
        const int64_t ps = (N / 4) | 1;
        const int64_t image_words = sort_lin ? ((64 * N * 4 + 16 + 1023) >> 10) * 256 : 64 * ps * 4;
        // one region per wave: votes image | tokens image | the cells' truth values (256 bytes)
        const int64_t region_words = image_words * (tok ? 2 : 1) + 64;
        const int64_t tail_words = (n_valid && B <= scv::kMaxSortedB ? ((B + 3) & ~3) : 0) + ((((int64_t)B * (nv + 1) + 1) & ~(int64_t)1) + 4 * (int64_t)B);
        int W = rk.waves;
        while (W > 1 && (W * region_words + tail_words) * 4 > ctx->lds_max) --W;
        // whole waves per SIMD: the steps are VALU-bound, so with 14 waves (N = 37 ... 44) two SIMDs carry 4 and set the pace of a round that 12 waves finish as
        // fast with the other two SIMDs idle a quarter of the time (N = 40 / 44 / 56: 72.1 / 72.7 / 82.4 -> 67.5 / 68.0 / 78.1 us, profiles/r06_sort_waves_ab.log);
        // not below 8: with tokens 7 / 6 / 5 waves beat 4 by 10-17 % (one wave per SIMD cannot hide its own copy)
        if (W > 8) W &= ~3;
        if ((W * region_words + tail_words) * 4 <= ctx->lds_max && W >= (rk.waves >= 16 ? 4 : 2)) {
            a.wave_lds_words = (int32_t)region_words;
            const size_t lds = (size_t)(W * region_words + tail_words) * sizeof(uint32_t);
            SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rk.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int64_t nsteps = (ncells + 63) / 64;
            // persistent: one workgroup per CU (or as many as the LDS lets be resident), 16 waves per CU at most
            int per_cu = (int)(ctx->lds_max / (int64_t)lds);
            if (per_cu < 1) per_cu = 1;
            if (per_cu * W > 16) per_cu = 16 / W > 0 ? 16 / W : 1;
            // (few steps: smaller workgroups on more CUs, scvote_dispatch.h)
            const int Wl = scv::spread_waves(nsteps, W, ctx->num_cus, nv <= 16 ? 8 : 4);
            const size_t lds_l = (size_t)(Wl * region_words + tail_words) * sizeof(uint32_t);
            int64_t grid = (nsteps + Wl - 1) / Wl;
            if (grid > (int64_t)ctx->num_cus * per_cu) grid = (int64_t)ctx->num_cus * per_cu;
            // cells per grid step a multiple of B: every lane slot then sees one budget and keeps its counters in registers
            if ((grid * Wl * 64) % B != 0 && grid > B) grid -= grid % B;
            if (ctx->grid_override > 0) grid = ctx->grid_override;
            ctx->stat_sort_cells += 1;
            if (ev) SCV_HIP(hipEventRecord(ev->a, ctx->stream));
            hipLaunchKernelGGL(rk.fn, dim3((unsigned)grid), dim3((unsigned)(Wl * 64)), lds_l, ctx->stream, a);
            SCV_HIP(hipGetLastError());
            return finish(ev);
        }
        kind = N <= ctx->tiny_n_max && lane_kernel_lds(B, N <= 4 ? 4 : (N <= 8 ? 8 : (N <= 16 ? 16 : 32))) <= (size_t)60 * 1024 ? LANE : REG;   // (hundreds of budgets: the LDS tables do not fit)
    }

    if (kind == REG) {
        // ---- register-resident cells: workgroups of independent waves, 8 KiB of LDS each
        const bool vec = rows_aligned;
        // shape of the kernel from the slots a row needs (N, or N + 3 for unaligned rows); the "reg_shape" option forces one for
        // A/B runs: g*100 + v (sparse: g lanes per cell, v vectors per lane) or 1000 + v*10 + h (dense: h parts of v vectors)
        int g = 0, v = 0, h = 0;
        if (Nreg <= 64) { g = 16; v = 1; }
        else if (Nreg <= 96) { g = 8; v = 3; }        // round 6: 96 slots for 65 ... 96 votes (8 lanes x 3 vectors, 8-bit bins), not 128 ("reg_shape" = 1602: round 5's shape)
        else if (Nreg <= 128) { g = 8; v = 4; }       // ... and 8 lanes x 4 vectors for 97 ... 128 (3-10 % over 16 lanes x 2: eight cells per wave share the fixed work)
        else if (Nreg <= 256) { g = 16; v = 4; }
        else if (Nreg <= 512) { g = 32; v = 4; }
        else if (Nreg <= 896) { g = 64; v = 4; }
        else if (Nreg <= 1024) { v = 4; h = 1; }      // round 3 (with the pivots): one dense part beats the sparse read-back from ~900 votes (N = 1024: 189 vs 200 us; 768: 224 vs 215)
        else if (Nreg <= 2048) { v = 4; h = 2; }      // 4 KiB parts: 126-136 VGPRs, 3 waves per SIMD (8 KiB parts: 200, 2 waves; measured 73 vs 79 us)
        else if (Nreg <= 4096) { v = 4; h = 4; }
        else { v = 4; h = 8; }                        // 4096 < N <= 8192: 4.5 -> 5.4 TB/s at N = 4608 against the streaming kernel, equal at 8192
        // (round 6: the dense codes are 1041 ... 1048 ONLY -- until now every value >= 1000 was read as a dense code, so 1601 / 1602 / 1604 / 3204 /
        //  6404 were never honoured: a forced sparse shape silently ran the auto choice)
        if (ctx->reg_shape >= 1040 && ctx->reg_shape < 1050) {
            const int fv = (ctx->reg_shape - 1000) / 10, fh = ctx->reg_shape % 10;
            if (fv == 4 && (fh == 1 || fh == 2 || fh == 4 || fh == 8) && (int64_t)256 * fv * fh >= Nreg) { g = 0; v = fv; h = fh; }
        } else if (ctx->reg_shape > 0) {
            const int fg = ctx->reg_shape / 100, fv = ctx->reg_shape % 100;
            const bool have = (fg == 16 && (fv == 1 || fv == 2 || fv == 4)) || ((fg == 32 || fg == 64) && fv == 4) || (fg == 8 && (fv == 3 || fv == 4));
            if (have && (int64_t)4 * fg * fv >= Nreg && (fg != 8 || scv::pick_reg_g8(fv, tok, vec).fn)) { g = fg; v = fv; h = 0; }   // ((8, 4): A/B builds only)
        }
        const int64_t cpw = h ? 1 : 64 / g;
        const int64_t nbatches = (ncells + cpw - 1) / cpw;
        a.wave_lds_words = (int32_t)(scv::kRegWaveWords16 + (n_valid && B <= scv::kMaxSortedB ? ((B + 3) & ~3) : 0));
        const RegKernel rk = h ? pick_dense_kernel(v, h, tok, vec) : pick_reg_kernel(g, v, tok, vec);
        KernelFn fn = rk.fn;
        // workgroup = all the waves of the shape a CU holds (16 / 12 / 8: the kernel's launch bounds), fewer when the
        // n_valid cache makes 16 regions overflow the LDS
        int WPG = rk.waves;
        while (WPG > 4 && (int64_t)WPG * a.wave_lds_words * 4 + 1024 > ctx->lds_max) WPG -= 4;
        size_t lds = (size_t)WPG * a.wave_lds_words * sizeof(uint32_t);
        SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // The grid is persistent, so it must not exceed what is resident at once (a surplus workgroup would start
        // only when another one has finished ALL its batches: measured 2x).  Ask the runtime.
        int per_cu = 0;
        SCV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), WPG * 64, lds));
        if (per_cu < 1) per_cu = 1;
        // Per-budget counters in the LDS the resident workgroups leave free (no occupancy lost): [B][TCL] tie classes +
        // 2 B sums per workgroup, flushed by the same launch -> no scv_reduce_cells launch, no cell scratch.  Tie
        // classes that do not fit (TCL < min(N, 1024) + 1; only with many budgets) go to memory directly.
        if (want_counters && !use_reduce) {
            const int64_t spare_words = (ctx->lds_max / per_cu - (int64_t)lds) / 4 - 64;
            const int64_t full = (N < 1024 ? N : 1024) + 1;
            int64_t tcl = spare_words > 4 * (int64_t)B + 2 ? (spare_words - 4 * (int64_t)B - 2) / B : 0;
            if (tcl > full) tcl = full;
            if (tcl >= 8 || tcl == full) {
                a.acc_classes = (int32_t)tcl;
                lds += ((((size_t)B * tcl + 1) & ~(size_t)1) + 4 * (size_t)B) * sizeof(uint32_t);
                SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                int again = 0;
                SCV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&again, reinterpret_cast<const void*>(fn), WPG * 64, lds));
                if (again >= 1 && again < per_cu) per_cu = again;   // (not expected: the region was sized from the spare LDS)
                ctx->stat_lds_counters += 1;
            }
        }
        int64_t grid = (int64_t)ctx->num_cus * per_cu;                     // workgroups of WPG independent waves
        if (ctx->grid_override > 0) grid = ctx->grid_override;
        const int64_t wgs_needed = (nbatches + WPG - 1) / WPG;
        if (grid >= wgs_needed) grid = wgs_needed;
        else {
            // a wave strides over the batches by grid * WPG: keep the stride coprime with B so that every wave
            // visits every budget (ragged n_valid would otherwise give some waves only the long budgets)
            auto gcd = [](int64_t x, int64_t y) { while (y) { const int64_t t = x % y; x = y; y = t; } return x; };
            while (grid > 1 && gcd(grid * WPG, B) != 1 && gcd(grid, B) != 1) --grid;
        }
        if (ev) SCV_HIP(hipEventRecord(ev->a, ctx->stream));
        hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(WPG * 64), lds, ctx->stream, a);
        SCV_HIP(hipGetLastError());
        return finish(ev);
    }

    if (kind == LANE && (N == 1 || N == 2 || N == 4) && !pool_rows && ncells < (1ll << 29) &&
        (((uintptr_t)answers & 15u) == 0) && (!tok || ((uintptr_t)tokens & 15u) == 0) && lane_kernel_lds(B, (int)N) <= (size_t)60 * 1024) {
        // ---- cells of exactly 1, 2 or 4 votes (the reference's most common sizes, o1.py:276,302): a block of 256 / 128 / 64 cells per wave and step
        const int64_t nblocks = ncells / (256 / N);                // whole blocks
        const int threads = 1024;
        const size_t lds = lane_kernel_lds(B, (int)N);
        int64_t grid = (nblocks + threads / 64 - 1) / (threads / 64);
        if (grid < 1) grid = 1;
        const int64_t cap = (int64_t)ctx->num_cus * 2;             // two workgroups per CU: 32 waves, each with a block in flight behind the one it counts
        if (grid > cap) grid = cap;
        // cells per grid step a multiple of B: every cell slot of a lane then keeps its budget and its counters stay in registers
        if ((grid * (threads / 64) * (256 / N)) % B != 0 && grid > B) grid -= grid % B;
        if (ctx->grid_override > 0) grid = ctx->grid_override;
        // N = 1 (o1.py:302, 276: the reference's most common call): a kernel whose loop body is one compare per cell (round 6), when the grid
        // step is a multiple of B (it is, unless the launch is smaller than B workgroups or the grid is forced); it also takes the cells behind the
        // last whole block (large launches only: a few hundred cells are the general kernel's)
        // (N = 2, o1.py:276 at T = 4096: scv_two_votes, the same idea with blocks of 128 cells)
        const bool one = N <= 2 && (grid * (threads / 64) * (256 / N)) % B == 0 && (size_t)B * 32 <= (size_t)60 * 1024 && (ncells % (256 / N) == 0 || ncells >= 65536);
        if (one || ncells % (256 / N) == 0) {
            ctx->stat_few_votes += 1;
            if (ev) SCV_HIP(hipEventRecord(ev->a, ctx->stream));
            if (one && N == 1) {
                const size_t lds1 = (size_t)B * 2 * sizeof(unsigned long long);
                if (tok) hipLaunchKernelGGL((scv::scv_one_vote<true>), dim3((unsigned)grid), dim3(threads), lds1, ctx->stream, a);
                else hipLaunchKernelGGL((scv::scv_one_vote<false>), dim3((unsigned)grid), dim3(threads), lds1, ctx->stream, a);
                ctx->stat_one_vote += 1;
            } else if (one) {
                const size_t lds2 = (size_t)B * 4 * sizeof(unsigned long long);
                if (tok) hipLaunchKernelGGL((scv::scv_two_votes<true>), dim3((unsigned)grid), dim3(threads), lds2, ctx->stream, a);
                else hipLaunchKernelGGL((scv::scv_two_votes<false>), dim3((unsigned)grid), dim3(threads), lds2, ctx->stream, a);
                ctx->stat_one_vote += 1;
            } else {
#define SCV_FEW(NVV) do { if (tok) hipLaunchKernelGGL((scv::scv_few_votes<NVV, true>), dim3((unsigned)grid), dim3(threads), lds, ctx->stream, a); \
                          else hipLaunchKernelGGL((scv::scv_few_votes<NVV, false>), dim3((unsigned)grid), dim3(threads), lds, ctx->stream, a); } while (0)
                if (N == 1) SCV_FEW(1); else if (N == 2) SCV_FEW(2); else SCV_FEW(4);
#undef SCV_FEW
            }
            SCV_HIP(hipGetLastError());
            return finish(ev);
        }
    }

    if (kind == LANE) {
        // ---- tiny cells, one lane per cell; counters accumulated in LDS and flushed by the same launch
        const int nv = N <= 4 ? 4 : (N <= 8 ? 8 : (N <= 16 ? 16 : 32));
        const int threads = (nv == 32 || (nv == 16 && tok)) ? 512 : 1024;      // register budget: 2 x nv votes (+ tokens) per lane
        const size_t lds = lane_kernel_lds(B, nv);
        a.wave_lds_words = rows_aligned ? 1 : 0;   // "vec" flag
        int64_t grid = (ncells + threads - 1) / threads;
        if (grid > ctx->num_cus) grid = ctx->num_cus;            // one workgroup per CU: the flush costs one atomic per workgroup and counter
        // grid * threads a multiple of B: every lane then sees one budget and keeps its counters in registers
        if ((grid * threads) % B != 0 && grid > B) grid -= grid % B;
        if (ctx->grid_override > 0) grid = ctx->grid_override;
        if (ev) SCV_HIP(hipEventRecord(ev->a, ctx->stream));
#define SCV_LANE(NVV, TT, TOKK) hipLaunchKernelGGL((scv::scv_lane_cells<NVV, TT, TOKK>), dim3((unsigned)grid), dim3(TT), lds, ctx->stream, a)
        if (nv == 4) { if (tok) SCV_LANE(4, 1024, true); else SCV_LANE(4, 1024, false); }
        else if (nv == 8) { if (tok) SCV_LANE(8, 1024, true); else SCV_LANE(8, 1024, false); }
        else if (nv == 16) { if (tok) SCV_LANE(16, 512, true); else SCV_LANE(16, 1024, false); }
        else SCV_LANE(32, 512, false);                              // (with tokens: REG, see the decision above)
#undef SCV_LANE
        SCV_HIP(hipGetLastError());
        return finish(ev);
    }

    // ---- streaming kernel geometry ----------------------------------------------------------------
    int copies = ctx->copies, threads = ctx->threads, wg_per_cu = ctx->wg_per_cu, unroll = ctx->unroll;
    if (!ctx->user_tuned) {
        // measured crossover (tools/crossover.py): the per-cell fold costs 1024*R LDS words, so short
        // cells want small R and several cells in flight per CU; long cells want one big workgroup.
        if (N < 4096) { copies = 4; threads = 256; wg_per_cu = 8; unroll = 2; }     // latency-bound: most cells in flight
        else if (N < 32768) { copies = 8; threads = 256; wg_per_cu = 4; unroll = 4; }
        else if (N < 262144) { copies = 16; threads = 512; wg_per_cu = 2; unroll = 4; }
        else { copies = 16; threads = 1024; wg_per_cu = 1; unroll = 4; }
        // fewer cells than CUs: one cell per CU whatever the band, so give each the widest workgroup
        if (ncells <= ctx->num_cus && N >= 4096) { copies = 16; threads = 1024; wg_per_cu = 1; unroll = 4; }
    }
    const size_t lds = ((size_t)scv::kBins * copies + scv::kRedWords + scv::kMaxSortedB) * sizeof(uint32_t);
    if ((int64_t)lds > ctx->lds_max) return fail(SCV_ERR_ARG, "LDS request %zu exceeds device limit %lld", lds, (long long)ctx->lds_max);
    const int by_lds = (int)((160 * 1024) / lds);
    const int by_waves = 2048 / threads;
    if (wg_per_cu > by_lds) wg_per_cu = by_lds;
    if (wg_per_cu > by_waves) wg_per_cu = by_waves;
    if (wg_per_cu < 1) wg_per_cu = 1;
    const int64_t slots = (int64_t)ctx->num_cus * wg_per_cu;

    // split-N when whole cells cannot fill the chip: S workgroups per cell, then one merge launch.
    // Measured (tools/segs_sweep.py): best is ONE round of items (at most one per workgroup slot) with
    // segments of at least 512 KiB -- more, smaller segments only add fold/publish/merge work.
    // Round 6: the segments of a cell are merged INSIDE the launch (device atomics into the cell's histogram in memory, the last segment to
    // arrive runs the epilogue), so a segment costs its fold + 16 atomic instructions, not 4 KiB of traffic and a share of a second kernel:
    // segments of 256 KiB, one round over every workgroup slot (a single cell of 2^24 votes: 128 x 512 KiB + merge launch = 27.8 us in round 5).
    int64_t S = 1;
    if ((ctx->path == 2 || (ctx->path == 0 && 2 * ncells <= slots && N * 4 >= (1 << 20))) && ncells <= scv::kSplitTickets) {
        if (ctx->segs_override > 0) S = ctx->segs_override;
        else {
            S = slots / ncells;
            const int64_t by_size = (N * 4) / (ctx->split_seg_kb << 10);
            if (S > by_size) S = by_size;
        }
        if (S > 4096) S = 4096;
        if (S < 1) S = 1;
    }
    if (S > 1) {
        a.segs = (int32_t)S;
        a.seg_len = ((N + S - 1) / S + 3) & ~(int64_t)3;     // multiple of 4 votes: segments start 16-byte aligned in aligned rows
        a.sorted = 0;                                        // split items are numbered cell-major
        const size_t hist_bytes = (size_t)ncells * scv::kBins * sizeof(uint32_t);
        if (int rc = ensure_partial(ctx, hist_bytes + (size_t)ncells * sizeof(long long) + 256)) return rc;
        a.partial = static_cast<uint32_t*>(ctx->d_partial);
        a.partial_tok = reinterpret_cast<long long*>(static_cast<char*>(ctx->d_partial) + ((hist_bytes + 255) / 256) * 256);
        a.tickets = static_cast<uint32_t*>(ctx->d_tickets);
    }
    // the single-launch epilogues exist for the geometries the library picks itself (scvote_dispatch.h)
    const bool have_xtra = pick_kernel(copies, threads, unroll, tok, true) != nullptr;
    if (a.overwrite && (S > 1 || !have_xtra)) {
        // split cells use the tickets for their own hand-off (kept simple: no second epilogue behind it), and a
        // hand-tuned geometry may have no epilogue variant: overwrite = memset node + accumulate here.
        a.overwrite = 0;
        overwrite_fused = false;
        if (S == 1) a.tickets = nullptr;
        a.ow_tie = a.ow_tok = a.ow_truth = nullptr;
        a.cells = cells; a.cell_tokens = cell_tokens;
        a.tie_hits = reinterpret_cast<unsigned long long*>(tie);
        a.token_sum = reinterpret_cast<unsigned long long*>(tok_sum);
        a.truth_sum = reinterpret_cast<unsigned long long*>(truth_sum);
        if (tie) SCV_HIP(hipMemsetAsync(tie, 0, (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t), ctx->stream));
        if (tok_sum) SCV_HIP(hipMemsetAsync(tok_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
        if (truth_sum) SCV_HIP(hipMemsetAsync(truth_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
    }
    if (overwrite_fused) ctx->stat_overwrite_fused += 1;                                     // counted once the form is final
    const int64_t nitems = ncells * S;
    int64_t grid = slots;
    if (ctx->grid_override > 0) grid = ctx->grid_override;
    if (grid > nitems) grid = nitems;
    if (ctx->grid_override <= 0) {
        // A 4 MiB cell is several percent of a launch: a ragged last round (10000 cells over 256
        // workgroups = 39.06 rounds) idles most of the chip for a whole cell.  Use the smallest grid
        // with the same number of rounds, so every workgroup streams `rounds` (or rounds-1) items.
        const int64_t rounds = (nitems + grid - 1) / grid;
        grid = (nitems + rounds - 1) / rounds;
    }

    // bootstrap inside this launch (scv_aggregate_bootstrap_i32): whole cells only, code table + counters in the
    // histogram's LDS, and the whole grid resident at once (it meets at a grid barrier)
    if (ctx->boot_req && ctx->boot_path <= 1 && S == 1 && a.cells && have_xtra) {
        const scv_ctx::BootReq& rq = *ctx->boot_req;
        const size_t need_words = (((size_t)B * rq.M + 3) & ~(size_t)3) + ((size_t)ncells + 1) / 2;
        KernelFn fx = pick_kernel(copies, threads, unroll, tok, true);
        int per_cu = 0;
        SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fx), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SCV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fx), threads, lds));
        if (need_words <= (size_t)scv::kBins * copies && grid <= (int64_t)per_cu * ctx->num_cus && P <= 0xFFFFFFFFll) {
            a.boot = 1;
            a.boot_r0 = rq.r0; a.boot_r1 = rq.r1; a.boot_M = rq.M; a.boot_seed = rq.seed;
            a.boot_spins = (uint32_t)(ctx->boot_spin_limit > 0 ? ctx->boot_spin_limit : 1);
            a.boot_out = reinterpret_cast<unsigned long long*>(rq.out);
            a.tickets = static_cast<uint32_t*>(ctx->d_tickets);
        }
    }
    const bool xtra = a.overwrite != 0 || a.boot != 0;
    KernelFn fn = pick_kernel(copies, threads, unroll, tok, xtra);
    if (!fn) return fail(SCV_ERR_ARG, "streaming geometry copies=%d threads=%d unroll=%d is not instantiated (scv_set_tuning lists the ones that are)", copies, threads, unroll);
    SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (ev) SCV_HIP(hipEventRecord(ev->a, ctx->stream));
    bool launched = false;
    if (a.boot) {
        // The workgroups meet at a grid barrier: ask the runtime for a COOPERATIVE launch, which only starts when the
        // whole grid can be resident at once whatever else (other streams, RCCL's kernels, another process) is on the
        // device.  Not available while the stream is being captured into a graph: there the ordinary launch is used and
        // the bounded spin + the repair in scv_sync cover a grid that turned out not to be co-resident.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (ctx->boot_path == 0 && cap == hipStreamCaptureStatusNone) {
            void* kargs[] = {const_cast<scv::AggArgs*>(&a)};
            const hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3((unsigned)grid), dim3((unsigned)threads), kargs, (unsigned)lds, ctx->stream);
            if (ce == hipSuccess) { launched = true; ctx->stat_boot_cooperative += 1; }
            else {
                // refused (grid too large for a cooperative launch right now, or no support): two kernels instead
                (void)hipGetLastError();
                a.boot = 0; a.boot_out = nullptr;
                fn = pick_kernel(copies, threads, unroll, tok, a.overwrite != 0);
                SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
        }
        if (a.boot) {
            const scv_ctx::BootReq& rq = *ctx->boot_req;
            ctx->boot_req->fused = true;
            ctx->stat_boot_fused += 1;
            ctx->boot_last.cells = a.cells; ctx->boot_last.P = P; ctx->boot_last.B = B; ctx->boot_last.r0 = rq.r0; ctx->boot_last.r1 = rq.r1;
            ctx->boot_last.M = rq.M; ctx->boot_last.seed = rq.seed; ctx->boot_last.out = rq.out; ctx->boot_last.valid = true;
        }
    }
    if (!launched) hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3((unsigned)threads), lds, ctx->stream, a);
    SCV_HIP(hipGetLastError());
    return finish(ev);
}

int launch_dense(scv_ctx* ctx, const int32_t* answers, const int32_t* tokens, const int32_t* n_valid,
                 const int32_t* truth, int64_t P, int32_t B, int64_t N, scv_cell* cells, int64_t* cell_tokens,
                 int64_t* tie, int64_t* tok_sum, int64_t* truth_sum) {
    return launch_aggregate(ctx, answers, tokens, n_valid, truth, P, B, N, cells, cell_tokens, tie, tok_sum, truth_sum, false);
}

// Short pools (N <= 4096: the reference's own sizes) go through the cell kernels: the pool row is re-read per budget, but out of
// the cache, and a cell costs what its n_valid votes cost -- the one-pass streaming kernel pays a workgroup-wide fold per
// boundary (measured at P x N = 10^5 x 256, budgets 1, 2, 4 ... N: 240 -> 60 us).
bool pool_rows_eligible(const scv_ctx* ctx, int32_t B, int64_t N, bool rows_aligned) {
    if (ctx->path != 0 || N < 1) return false;
    if (N > ctx->tiny_n_max) return N <= ctx->reg_n_max && N + (rows_aligned ? 0 : 3) <= 4096;   // (+ 3: an unaligned pool row is read as its aligned superset)
    const int nv = N <= 4 ? 4 : (N <= 8 ? 8 : (N <= 16 ? 16 : 32));
    return lane_kernel_lds(B, nv) <= (size_t)60 * 1024;
}

// pools of up to 64 samples: scv_lane_prefix (tie classes 0..nv, two sums, order + sorted n_valid per budget in LDS; the snapshots of a
// wave's 64 x B cells staged in LDS: 16 bytes each, 24 with tokens -- budget lists whose snapshots do not fit even a 256-thread workgroup
// run on scv_prefix_pool, which holds one record per lane whatever the number of budgets)
bool prefix_lane_eligible(const scv_ctx* ctx, int32_t B, int64_t N, bool tok, int* nv, size_t* lds, int* threads) {
    if (ctx->path != 0 || N < 1 || N > 64 || B > scv::kMaxSortedB) return false;
    *nv = N <= 4 ? 4 : (N <= 8 ? 8 : (N <= 16 ? 16 : (N <= 32 ? 32 : 64)));
    *lds = ((((size_t)B * (*nv + 1) + 1) & ~(size_t)1) + 6 * (size_t)B) * sizeof(uint32_t);
    if (*lds > (size_t)24 * 1024) return false;
    const size_t counters_bytes = (*lds + 15) & ~(size_t)15;
    const size_t per_wave = (size_t)64 * B * (sizeof(scv_cell) + (tok ? sizeof(int64_t) : 0));
    for (int t = *nv == 64 ? 256 : 1024; t >= 256; t >>= 1) {
        if (counters_bytes + (size_t)(t / 64) * per_wave + 1024 <= (size_t)ctx->lds_max) { *threads = t; return true; }
    }
    return false;
}

// One pass per problem over its pool row, every budget a snapshot of the running mode statistics (scvote_prefix.hip.h).
int launch_prefix_pool(scv_ctx* ctx, const int32_t* pool, const int32_t* tokens, const int32_t* n_valid, const int32_t* truth,
                       int64_t P, int32_t B, int64_t N, scv_cell* cells, int64_t* cell_tokens, int64_t* tie, int64_t* tok_sum,
                       int64_t* truth_sum, bool rows_aligned, int skip_sortable = 0, EventPair* ev_open = nullptr, bool counters_cleared = false) {
    scv::AggArgs a;
    a.skip_sortable = skip_sortable;
    a.pool_rows = 1;
    a.answers = pool; a.tokens = tokens; a.n_valid = n_valid; a.truth = truth;
    a.ncells = P * (int64_t)B; a.N = N; a.B = B; a.P = P;
    a.cells = cells; a.cell_tokens = cell_tokens;
    a.tie_hits = reinterpret_cast<unsigned long long*>(tie);
    a.token_sum = reinterpret_cast<unsigned long long*>(tok_sum);
    a.truth_sum = reinterpret_cast<unsigned long long*>(truth_sum);
    a.err_flag = ctx->d_err;
    a.prefetch = 0; a.sorted = 1;
    a.segs = 1; a.seg_len = N; a.partial = nullptr; a.partial_tok = nullptr; a.wave_lds_words = 0; a.acc_classes = 0;
    a.tickets = nullptr; a.overwrite = 0; a.ow_tie = a.ow_tok = a.ow_truth = nullptr; a.boot = 0; a.boot_r0 = a.boot_r1 = 0; a.boot_M = 1; a.boot_spins = 0; a.boot_seed = 0; a.boot_out = nullptr;
    const bool tok = tokens != nullptr;
    const bool want_counters = tie || truth_sum || (tok && tok_sum);
    // lanes per problem: 16 up to 1024 votes (4 problems per wave share the fixed work per boundary; 16-bit bins: 16 waves per CU), 32 beyond
    // (measured: 1024 votes 68 us at 16 lanes against 82 at 32, 4096 votes 100 against 87)
    int g = N <= 1024 ? 16 : 32;
    if (ctx->reg_shape == 16 || ctx->reg_shape == 32) g = ctx->reg_shape;
    if (!rows_aligned) g = 16;                                       // (the only unaligned shape: scvote_prefix.hip)
    const RegKernel rk = pick_prefix_pool_kernel(g, tok, rows_aligned);
    a.wave_lds_words = (int32_t)(scv::prefix_pool_hist_words(g) + scv::kPrefixPoolLaneWords);
    int W = rk.waves;
    const size_t fixed_words = 2 * (size_t)B + scv::kPrefixPoolFixedWords;
    while (W > 1 && ((size_t)W * a.wave_lds_words + fixed_words + 64) * 4 > (size_t)ctx->lds_max) --W;
    size_t lds = ((size_t)W * a.wave_lds_words + fixed_words) * sizeof(uint32_t);
    const bool use_reduce = want_counters && ctx->fused_counters_max == 0;          // forced (tests): counters from the cell table
    if (use_reduce) {
        a.tie_hits = nullptr; a.token_sum = nullptr; a.truth_sum = nullptr;
        if (!a.cells || (tok && tok_sum && !a.cell_tokens)) {
            const size_t cb = (size_t)a.ncells * sizeof(scv_cell);
            if (int rc = ensure_cells(ctx, cb + (size_t)a.ncells * sizeof(int64_t) + 256)) return rc;
            if (!a.cells) a.cells = static_cast<scv_cell*>(ctx->d_cells);
            if (tok && !a.cell_tokens) a.cell_tokens = reinterpret_cast<int64_t*>(static_cast<char*>(ctx->d_cells) + ((cb + 255) / 256) * 256);
        }
    }
    if (ctx->overwrite_counters && want_counters && !counters_cleared) {      // overwrite semantics: a memset node in front
        if (tie) SCV_HIP(hipMemsetAsync(tie, 0, (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t), ctx->stream));
        if (tok_sum) SCV_HIP(hipMemsetAsync(tok_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
        if (truth_sum) SCV_HIP(hipMemsetAsync(truth_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
    }
    // per-budget counters in the LDS the histograms leave free: [B][TCL] tie classes + 2 B sums per workgroup, flushed by the same launch
    if (want_counters && !use_reduce) {
        const int64_t spare_words = ((int64_t)ctx->lds_max - (int64_t)lds) / 4 - 64;
        const int64_t full = (N < 1024 ? N : 1024) + 1;
        int64_t tcl = spare_words > 4 * (int64_t)B + 2 ? (spare_words - 4 * (int64_t)B - 2) / B : 0;
        if (tcl > full) tcl = full;
        if (tcl >= 8 || tcl == full) {
            a.acc_classes = (int32_t)tcl;
            lds += ((((size_t)B * tcl + 1) & ~(size_t)1) + 4 * (size_t)B) * sizeof(uint32_t);
            ctx->stat_lds_counters += 1;
        }
    }
    SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rk.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    SCV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(rk.fn), W * 64, lds));
    if (per_cu < 1) per_cu = 1;
    const int64_t nbatches = (P + (64 / g) - 1) / (64 / g);
    int64_t grid = (int64_t)ctx->num_cus * per_cu;
    if (ctx->grid_override > 0) grid = ctx->grid_override;
    const int64_t wgs_needed = (nbatches + W - 1) / W;
    if (grid > wgs_needed) grid = wgs_needed;
    EventPair* ev = ev_open;                             // (open: scv_sort_prefix was queued in front, inside the same pair)
    if (!ev_open) {
        if (int rc = next_event_pair(ctx, &ev)) return rc;
        if (ev) SCV_HIP(hipEventRecord(ev->a, ctx->stream));
    }
    hipLaunchKernelGGL(rk.fn, dim3((unsigned)grid), dim3((unsigned)(W * 64)), lds, ctx->stream, a);
    SCV_HIP(hipGetLastError());
    ctx->stat_prefix_pool += 1;
    if (use_reduce) {
        int64_t chunks = (P + 2047) / 2048;
        const int64_t cap = ((int64_t)ctx->num_cus * 8 + B - 1) / B;
        if (chunks > cap) chunks = cap;
        if (chunks < 1) chunks = 1;
        auto* th = reinterpret_cast<unsigned long long*>(tie);
        auto* ts = reinterpret_cast<unsigned long long*>(tok_sum);
        auto* tc = reinterpret_cast<unsigned long long*>(truth_sum);
        if (tok && tok_sum) hipLaunchKernelGGL((scv::scv_reduce_cells<true>), dim3((unsigned)chunks, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, ctx->stream, a.cells, a.cell_tokens, P, B, th, ts, tc);
        else hipLaunchKernelGGL((scv::scv_reduce_cells<false>), dim3((unsigned)chunks, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, ctx->stream, a.cells, a.cell_tokens, P, B, th, ts, tc);
        SCV_HIP(hipGetLastError());
    }
    if (ev) SCV_HIP(hipEventRecord(ev->b, ctx->stream));
    ctx->err_dirty = true;
    return SCV_OK;
}

// Budgets that are powers of two (and the whole row) over pools of 17 .. 128 votes: every budget out of ONE sort per problem
// (scv_sort_prefix, scvote_sort_prefix.hip.h).  Returns through *queued whether the kernel was launched; the kernel itself leaves without
// side effects when some budget is not of that form (the caller then queues the general kernel behind it, a.skip_sortable = *nv_out).
int launch_sort_prefix(scv_ctx* ctx, const int32_t* pool, const int32_t* tokens, const int32_t* n_valid, const int32_t* truth,
                       int64_t P, int32_t B, int64_t N, scv_cell* cells, int64_t* cell_tokens, int64_t* tie, int64_t* tok_sum,
                       int64_t* truth_sum, bool promised, bool* queued, int* nv_out) {
    *queued = false;
    const int nv = N <= 32 ? 32 : (N <= 64 ? 64 : 128);
    // pools of 68 .. 128 votes WITH tokens (the reference's largest pool, o1.py:266-276, and its drop-in always sums tokens, o1.py:195): scv_sort_prefix2<true> --
    // the votes as without tokens, the token rows in token steps of their own that the waves without a sort step in the last, partial round take first
    // (round 6; scvote_sort_prefix.hip.h)
    const bool tok = tokens != nullptr;
    const RegKernel rk = pick_sort_prefix_kernel(nv, tok);
    const int64_t ps = nv == 128 ? 17 : ((N / 4) | 1);                 // (128: the row goes through the image in two halves of up to 64 votes)
    // one image per wave (scv_sort_prefix: a step's tokens follow its votes through it)
    const int64_t region_words = 64 * ps * 4 + 64;
    const int64_t tail_words = scv::sort_prefix_tail_words(nv, B);
    int W = rk.waves;
    while (W > 2 && (W * region_words + tail_words) * 4 + 1024 > ctx->lds_max) --W;
    if ((W * region_words + tail_words) * 4 + 1024 > ctx->lds_max) return SCV_OK;
    scv::AggArgs a;
    a.pool_rows = 1;
    a.answers = pool; a.tokens = tok ? tokens : nullptr; a.n_valid = n_valid; a.truth = truth;
    a.ncells = P * (int64_t)B; a.N = N; a.B = B; a.P = P;
    a.cells = cells; a.cell_tokens = tok ? cell_tokens : nullptr;
    a.tie_hits = reinterpret_cast<unsigned long long*>(tie);
    a.token_sum = tok ? reinterpret_cast<unsigned long long*>(tok_sum) : nullptr;
    a.truth_sum = reinterpret_cast<unsigned long long*>(truth_sum);
    a.err_flag = ctx->d_err;
    a.prefetch = 0; a.sorted = 1;
    a.budgets_promised = promised ? 1 : 0;     // the list is KNOWN to be of the served form (promised by the caller: option prefix_path = 5, or read by a HOST-mode call)
    a.segs = 1; a.seg_len = N; a.partial = nullptr; a.partial_tok = nullptr; a.acc_classes = 0;
    a.tickets = nullptr; a.overwrite = 0; a.ow_tie = a.ow_tok = a.ow_truth = nullptr; a.boot = 0; a.boot_r0 = a.boot_r1 = 0; a.boot_M = 1; a.boot_spins = 0; a.boot_seed = 0; a.boot_out = nullptr;
    a.wave_lds_words = (int32_t)region_words;
    const size_t lds = (size_t)(W * region_words + tail_words) * sizeof(uint32_t);
    SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rk.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    SCV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(rk.fn), W * 64, lds));
    if (per_cu < 1) per_cu = 1;
    const int64_t nsteps = (P + 63) / 64;
    // fewer steps than the chip has wave slots: every CU gets a workgroup before any CU gets a second wave per SIMD (a step is ~1400 VALU instructions
    // per wave; two waves of a SIMD in the same step take turns: 4.9 against 2.6 us for the sort of a one-step launch)
    const int Wl = scv::spread_waves(nsteps, W, ctx->num_cus);
    const size_t lds_l = (size_t)(Wl * region_words + tail_words) * sizeof(uint32_t);
    int64_t grid = (nsteps + Wl - 1) / Wl;
    if (grid > (int64_t)ctx->num_cus * per_cu) grid = (int64_t)ctx->num_cus * per_cu;
    if (ctx->grid_override > 0) grid = ctx->grid_override;
    hipLaunchKernelGGL(rk.fn, dim3((unsigned)grid), dim3((unsigned)(Wl * 64)), lds_l, ctx->stream, a);
    SCV_HIP(hipGetLastError());
    ctx->stat_prefix_sort += 1;
    if (tok && nv == 128 && (cell_tokens || tok_sum)) ctx->stat_prefix_tokens += 1;   // (launches with the token steps of scv_sort_prefix2<true>)
    ctx->err_dirty = true;
    *queued = true;
    *nv_out = nv;
    return SCV_OK;
}

// Prefix budgets over one pool [P, N] (scv_aggregate_prefix_i32).  Same outputs as launch_aggregate on the dense [P, B, N]
// expansion.  Option "prefix_path" forces one of the three forms (parity tests).
int launch_prefix(scv_ctx* ctx, const int32_t* pool, const int32_t* tokens, const int32_t* n_valid,
                  const int32_t* truth, int64_t P, int32_t B, int64_t N, scv_cell* cells, int64_t* cell_tokens,
                  int64_t* tie, int64_t* tok_sum, int64_t* truth_sum) {
    const int64_t ncells = P * (int64_t)B;
    if (ncells == 0) return SCV_OK;
    int lane_nv = 0, lane_threads = 0;
    size_t lane_lds = 0;
    const bool lane_ok = prefix_lane_eligible(ctx, B, N, tokens != nullptr, &lane_nv, &lane_lds, &lane_threads) && (ctx->prefix_path == 0 || ctx->prefix_path == 1 || ctx->prefix_path == 5);
    const bool rows_aligned = (N % 4 == 0) && (((uintptr_t)pool & 15u) == 0) && (!tokens || ((uintptr_t)tokens & 15u) == 0);
    // pools of 17 .. 128 votes, budgets that are powers of two (the reference's own, o1.py:274-277): every budget out of one sort per problem.
    // The budgets live in n_valid: a HOST-mode call reads them; a DEVICE-mode call queues scv_sort_prefix AND the general kernel -- each
    // decides from n_valid, in its first microsecond, whether the launch is its own.
    int skip_sortable = 0;
    EventPair* ev_open = nullptr;
    bool counters_cleared = false;
    const bool want_any_counters = tie || truth_sum || (tokens && tok_sum);
    if (ctx->path == 0 && (ctx->prefix_path == 0 || ctx->prefix_path == 5) && ctx->sort_n_max >= 64 && ctx->fused_counters_max != 0 && rows_aligned &&
        N > 16 && N <= 128 && B <= scv::kMaxSortedB &&
        // pools of 68 .. 128 votes (scv_sort_prefix2: two sorts, a merge and a 128-vote scan per step) pay ~27 us for a launch of one step per wave (36 before round 6's 4-wave workgroups):
        // measured against scv_prefix_pool it wins from ~5e4 pools (4.9e4: 28.9 against 29.5 us, 6.6e4: 30.1 against 35.2, 2e5: 48 .. 63 against 82; 3e4: 153 against 69 in HOST-mode
        // chunks); with tokens (round 6) the sums come from token steps of the same launch (round 5 tried a second image: 114 against 114 us,
        // and left such calls on scv_prefix_pool); prefix_path = 5 selects it for any number of pools
        (N <= 64 || P >= 57344 || ctx->prefix_path == 5)) {
        // prefix_path = 5: the caller PROMISES budgets of that form (a DEVICE-mode call then queues scv_sort_prefix alone; a list that breaks
        // the promise is an error, reported like a domain error at the next synchronisation)
        const bool promised = ctx->prefix_path == 5;
        bool known = promised, served = true;
        if (ctx->nv_host) {
            known = true;
            const int64_t np = N <= 32 ? 16 : (N <= 64 ? 32 : 64);
            for (int32_t b = 0; b < B; ++b) {
                const int64_t n = ctx->nv_host[b];
                if (!(n <= 0 || n >= N || ((n & (n - 1)) == 0 && n <= np))) { served = false; break; }
            }
        }
        if (!served && promised) return fail(SCV_ERR_ARG, "prefix_path = 5 promises budgets that are 0, a power of two <= %d, or >= N", N <= 32 ? 16 : (N <= 64 ? 32 : 64));
        if (served) {
            if (int rc = next_event_pair(ctx, &ev_open)) return rc;
            if (ctx->overwrite_counters && want_any_counters) {
                if (tie) SCV_HIP(hipMemsetAsync(tie, 0, (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t), ctx->stream));
                if (tok_sum) SCV_HIP(hipMemsetAsync(tok_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
                if (truth_sum) SCV_HIP(hipMemsetAsync(truth_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
                counters_cleared = true;
            }
            if (ev_open) SCV_HIP(hipEventRecord(ev_open->a, ctx->stream));
            bool queued = false;
            if (int rc = launch_sort_prefix(ctx, pool, tokens, n_valid, truth, P, B, N, cells, cell_tokens, tie, tok_sum, truth_sum, known, &queued, &skip_sortable)) return rc;
            if (queued && known) {
                if (ev_open) SCV_HIP(hipEventRecord(ev_open->b, ctx->stream));
                return SCV_OK;
            }
            if (!queued) skip_sortable = 0;
        }
    }
    // pools of 65 .. 4096 votes (and shorter ones when forced): ONE pass per problem, every budget a snapshot (scv_prefix_pool)
    const bool pool_ok = ctx->path == 0 && N >= 1 && N <= 4096 && B <= scv::kMaxSortedB && (((ctx->prefix_path == 0 || ctx->prefix_path == 5) && !lane_ok) || ctx->prefix_path == 4);
    if (pool_ok) return launch_prefix_pool(ctx, pool, tokens, n_valid, truth, P, B, N, cells, cell_tokens, tie, tok_sum, truth_sum, rows_aligned, skip_sortable, ev_open, counters_cleared);
    if (!lane_ok && !skip_sortable && pool_rows_eligible(ctx, B, N, rows_aligned) && (ctx->prefix_path == 0 || ctx->prefix_path == 2)) {
        ctx->stat_prefix_cells += 1;
        return launch_aggregate(ctx, pool, tokens, n_valid, truth, P, B, N, cells, cell_tokens, tie, tok_sum, truth_sum, true);
    }
    scv::AggArgs a;
    a.skip_sortable = skip_sortable;
    a.pool_rows = 0;
    a.answers = pool; a.tokens = tokens; a.n_valid = n_valid; a.truth = truth;
    a.ncells = ncells; a.N = N; a.B = B; a.P = P;
    a.cells = cells; a.cell_tokens = cell_tokens;
    a.tie_hits = reinterpret_cast<unsigned long long*>(tie);
    a.token_sum = reinterpret_cast<unsigned long long*>(tok_sum);
    a.truth_sum = reinterpret_cast<unsigned long long*>(truth_sum);
    a.err_flag = ctx->d_err;
    a.prefetch = 0; a.sorted = 1;
    a.segs = 1; a.seg_len = N; a.partial = nullptr; a.partial_tok = nullptr; a.wave_lds_words = 0; a.acc_classes = 0;
    a.tickets = nullptr; a.overwrite = 0; a.ow_tie = a.ow_tok = a.ow_truth = nullptr; a.boot = 0; a.boot_r0 = a.boot_r1 = 0; a.boot_M = 1; a.boot_spins = 0; a.boot_seed = 0; a.boot_out = nullptr;
    const bool tok = tokens != nullptr;
    const bool want_counters = tie || truth_sum || (tok && tok_sum);
    // pools of up to 64 samples: one lane per problem, every budget out of one pass (scv_lane_prefix); its counters
    // come out of the same launch
    const bool use_reduce = want_counters && !lane_ok && (ctx->fused_counters_max == 0 || (ncells > ctx->fused_counters_max && N * (int64_t)B < (1 << 20)));
    if (use_reduce) {
        a.tie_hits = nullptr; a.token_sum = nullptr; a.truth_sum = nullptr;
        if (!a.cells || (tok && tok_sum && !a.cell_tokens)) {
            const size_t cb = (size_t)ncells * sizeof(scv_cell);
            if (int rc = ensure_cells(ctx, cb + (size_t)ncells * sizeof(int64_t) + 256)) return rc;
            if (!a.cells) a.cells = static_cast<scv_cell*>(ctx->d_cells);
            if (tok && !a.cell_tokens) a.cell_tokens = reinterpret_cast<int64_t*>(static_cast<char*>(ctx->d_cells) + ((cb + 255) / 256) * 256);
        }
    }
    if (ctx->overwrite_counters && want_counters && !counters_cleared) {      // overwrite semantics: a memset node in front (no fused variant here)
        if (tie) SCV_HIP(hipMemsetAsync(tie, 0, (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t), ctx->stream));
        if (tok_sum) SCV_HIP(hipMemsetAsync(tok_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
        if (truth_sum) SCV_HIP(hipMemsetAsync(truth_sum, 0, (size_t)B * sizeof(int64_t), ctx->stream));
    }
    EventPair* ev = ev_open;
    if (!ev_open) {
        if (int rc = next_event_pair(ctx, &ev)) return rc;
        if (ev) SCV_HIP(hipEventRecord(ev->a, ctx->stream));
    }
    if (lane_ok) {
        a.wave_lds_words = rows_aligned ? 1 : 0;   // "vec" flag
        // Workgroup size.  N <= 32: 1024 threads, one workgroup per CU (the end-of-launch flush is one device atomic per
        // workgroup and counter, ~12 ns each on one address).  N = 64 needs 95-151 VGPRs: 256 threads, as many workgroups
        // as are resident (measured 61-64 us against 84-86 us with 768 / 1024 threads, which spill or leave a second, nearly
        // empty round).  Snapshots are staged in LDS (24 bytes x 64 x B per wave with tokens), in smaller workgroups if need be
        // (prefix_lane_eligible picked the size).
        const size_t counters_bytes = (lane_lds + 15) & ~(size_t)15;
        const size_t per_wave = (size_t)64 * B * (sizeof(scv_cell) + (tok ? sizeof(int64_t) : 0));
        const int T = lane_threads;
        const size_t lds_total = counters_bytes + (size_t)(T / 64) * per_wave;
        KernelFn fn;
#define SCV_LP2(NVV, TBB) (tok ? (KernelFn)scv::scv_lane_prefix<NVV, TBB, true> : (KernelFn)scv::scv_lane_prefix<NVV, TBB, false>)
        fn = lane_nv == 4 ? SCV_LP2(4, 1024) : (lane_nv == 8 ? SCV_LP2(8, 1024) : (lane_nv == 16 ? SCV_LP2(16, 1024) : (lane_nv == 32 ? SCV_LP2(32, 1024) : SCV_LP2(64, 256))));
#undef SCV_LP2
        SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));
        int per_cu = 0;
        SCV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), T, lds_total));
        if (per_cu < 1 || T == 1024) per_cu = 1;
        int64_t grid = (P + T - 1) / T;
        if (grid > (int64_t)ctx->num_cus * per_cu) grid = (int64_t)ctx->num_cus * per_cu;
        if (ctx->grid_override > 0) grid = ctx->grid_override;
        hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3((unsigned)T), lds_total, ctx->stream, a);
        ctx->stat_prefix_lane += 1;
    } else {
        // one pass over the pool, a snapshot of the LDS histogram at every boundary (scv_prefix_hist): pools longer than 4096 votes
        int copies, threads, wg_per_cu;
        if (N < 32768) { copies = 8; threads = 256; wg_per_cu = 4; }
        else if (N < 262144) { copies = 16; threads = 512; wg_per_cu = 2; }
        else { copies = 16; threads = 1024; wg_per_cu = 1; }
        const size_t lds = ((size_t)scv::kBins * copies + scv::kRedWords + 2 * (size_t)scv::kMaxSortedB) * sizeof(uint32_t);
        int64_t grid = (int64_t)ctx->num_cus * wg_per_cu;
        if (grid > P) grid = P;
        { const int64_t rounds = (P + grid - 1) / grid; grid = (P + rounds - 1) / rounds; }
        KernelFn fn;
        if (threads == 256) fn = tok ? (KernelFn)scv::scv_prefix_hist<3, 256, 4, true> : (KernelFn)scv::scv_prefix_hist<3, 256, 4, false>;
        else if (threads == 512) fn = tok ? (KernelFn)scv::scv_prefix_hist<4, 512, 4, true> : (KernelFn)scv::scv_prefix_hist<4, 512, 4, false>;
        else fn = tok ? (KernelFn)scv::scv_prefix_hist<4, 1024, 4, true> : (KernelFn)scv::scv_prefix_hist<4, 1024, 4, false>;
        SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3((unsigned)threads), lds, ctx->stream, a);
    }
    SCV_HIP(hipGetLastError());
    if (use_reduce) {
        int64_t chunks = (P + 2047) / 2048;
        const int64_t cap = ((int64_t)ctx->num_cus * 8 + B - 1) / B;
        if (chunks > cap) chunks = cap;
        if (chunks < 1) chunks = 1;
        auto* th = reinterpret_cast<unsigned long long*>(tie);
        auto* ts = reinterpret_cast<unsigned long long*>(tok_sum);
        auto* tc = reinterpret_cast<unsigned long long*>(truth_sum);
        if (tok && tok_sum) hipLaunchKernelGGL((scv::scv_reduce_cells<true>), dim3((unsigned)chunks, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, ctx->stream, a.cells, a.cell_tokens, P, B, th, ts, tc);
        else hipLaunchKernelGGL((scv::scv_reduce_cells<false>), dim3((unsigned)chunks, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, ctx->stream, a.cells, a.cell_tokens, P, B, th, ts, tc);
        SCV_HIP(hipGetLastError());
    }
    if (ev) SCV_HIP(hipEventRecord(ev->b, ctx->stream));
    ctx->err_dirty = true;
    return SCV_OK;
}

// Read and clear the device error word (stream must be idle).
int fetch_err(scv_ctx* ctx, uint32_t* out, bool force = false) {
    *out = 0;
    if (!ctx->err_dirty && !force) return SCV_OK;
    SCV_HIP(hipMemcpyAsync(out, ctx->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SCV_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(uint32_t), ctx->stream));
    SCV_HIP(hipStreamSynchronize(ctx->stream));
    ctx->err_dirty = false;
    return SCV_OK;
}

int recover_fused_bootstrap(scv_ctx* ctx, uint32_t* w);

int check_err_word(scv_ctx* ctx, uint32_t w) {
    if (w & 4u) {
        // A workgroup of the fused vote + bootstrap launch gave up at the grid barrier (the grid was not co-resident:
        // only possible for the non-cooperative form, e.g. inside a captured graph).  The vote part is complete -- every
        // workgroup writes its cells and counters BEFORE it arrives at the barrier -- so the evaluation is repaired here:
        // barrier state reset, the whole bootstrap re-run as a separate launch over the same cell table, bit cleared.
        if (int rc = recover_fused_bootstrap(ctx, &w)) return rc;
    }
    if ((w & 1u) && !(ctx->flags & SCV_FLAG_CLAMP_TO_INVALID_BIN))
        return fail(SCV_ERR_DOMAIN, "a vote outside bins 0..1023 was seen; results are invalid");
    if (w & 2u) return fail(SCV_ERR_ARG, "bootstrap: a drawn hit had n_modes >= M");
    if (w & 8u) return fail(SCV_ERR_ARG, "prefix_path = 5 promised budgets that are 0, a power of two or >= N: the list in n_valid is not; nothing was computed");
    return SCV_OK;
}

int recover_fused_bootstrap(scv_ctx* ctx, uint32_t* w) {
    SCV_HIP(hipMemsetAsync(ctx->d_tickets, 0, 4 * sizeof(uint32_t), ctx->stream));      // arrivals / generation: a clean barrier again
    if (!ctx->boot_last.valid)
        return fail(SCV_ERR_ARG, "fused bootstrap: grid barrier timed out and the request is not known any more (graph replay of an older capture?); use option boot_path = 2");
    const scv_ctx::BootLast q = ctx->boot_last;
    *w &= ~(4u | 2u);                            // bit 1 (value 2: class overflow) is re-derived by the re-run over the complete table
    if (int rc = scv_bootstrap(ctx, q.cells, q.P, q.B, q.r0, q.r1, q.seed, q.M, SCV_MEM_DEVICE, q.out)) return rc;
    uint32_t again = 0;
    if (int rc = fetch_err(ctx, &again, true)) return rc;        // synchronises the stream
    *w |= again & ~4u;
    ctx->stat_boot_recovered += 1;
    return SCV_OK;
}

int ensure_stage(scv_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->d_stage_bytes) return SCV_OK;
    if (ctx->d_stage) { SCV_HIP(hipFree(ctx->d_stage)); ctx->d_stage = nullptr; ctx->d_stage_bytes = 0; }
    SCV_HIP(hipMalloc(&ctx->d_stage, bytes));
    ctx->d_stage_bytes = bytes;
    return SCV_OK;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" {

const char* scv_last_error(void) { return g_err; }
#ifdef SCV_TEST_HOOKS
const char* scv_version(void) { return "scvote 0.2 (gfx950) +testhooks"; }
#else
const char* scv_version(void) { return "scvote 0.2 (gfx950)"; }
#endif

int scv_device_count(void) {
    return guarded([&]() -> int {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) return 0;
        return n;
    });
}

int scv_create(scv_ctx** out, int device, uint32_t flags) {
    return guarded([&]() -> int {
        if (!out) return fail(SCV_ERR_ARG, "scv_create: out is NULL");
        *out = nullptr;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SCV_ERR_NO_DEVICE, "no HIP device visible");
        if (device < 0) { SCV_HIP(hipGetDevice(&device)); }
        if (device >= n) return fail(SCV_ERR_ARG, "device %d out of range (%d visible)", device, n);
        scv_ctx* ctx = new (std::nothrow) scv_ctx();
        if (!ctx) return fail(SCV_ERR_ALLOC, "out of host memory");
        ctx->device = device;
        ctx->flags = flags;
        DeviceGuard guard_;                       // restores the caller's current device on every return path
        hipError_t e = guard_.enter(device) == SCV_OK ? hipSuccess : hipErrorInvalidDevice;
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e == hipSuccess) { ctx->own_stream = true; e = hipMalloc((void**)&ctx->d_err, 256); }
        if (e == hipSuccess) e = hipMemset(ctx->d_err, 0, 256);
        if (e == hipSuccess) e = hipMalloc(&ctx->d_tickets, kTicketWords * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(ctx->d_tickets, 0, kTicketWords * sizeof(uint32_t));
        if (e == hipSuccess) { ctx->d_tickets_words = kTicketWords; e = hipDeviceSynchronize(); }
        if (e != hipSuccess) {
            int code = fail(-(int)e, "scv_create: %s", hipGetErrorString(e));
            if (ctx->d_tickets) (void)hipFree(ctx->d_tickets);
            if (ctx->d_err) (void)hipFree(ctx->d_err);
            if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return code;
        }
        ctx->num_cus = prop.multiProcessorCount;
        ctx->lds_max = 160 * 1024;  // gfx950: a single workgroup may declare all 160 KiB
        ctx->clock_khz = prop.clockRate;
        ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
        *out = ctx;
        return SCV_OK;
    });
}

int scv_destroy(scv_ctx* ctx) {
    return guarded([&]() -> int {
        if (!ctx) return SCV_OK;
        DeviceGuard guard_;
        (void)guard_.enter(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
        if (ctx->pipe) { ctx->pipe->shutdown(); delete ctx->pipe; ctx->pipe = nullptr; }
        if (ctx->d_stage) (void)hipFree(ctx->d_stage);
        if (ctx->small_h) (void)hipHostFree(ctx->small_h);
        if (ctx->small_d) (void)hipFree(ctx->small_d);
        if (ctx->d_partial) (void)hipFree(ctx->d_partial);
        if (ctx->d_tickets) (void)hipFree(ctx->d_tickets);
        if (ctx->d_cells) (void)hipFree(ctx->d_cells);
        if (ctx->d_err) (void)hipFree(ctx->d_err);
        if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return SCV_OK;
    });
}

int scv_set_stream(scv_ctx* ctx, void* hip_stream) {
    return guarded([&]() -> int {
        if (!ctx) return fail(SCV_ERR_ARG, "ctx is NULL");
        SCV_ENTER(ctx);
        // No synchronisation here (same contract as any set-stream call: ordering between the old and the
        // new stream is the caller's): a sync would be illegal while the new stream is being captured
        // into a hipGraph.  Only the ctx's own private stream is drained before it is destroyed.
        if (ctx->own_stream) {
            SCV_HIP(hipStreamSynchronize(ctx->stream));
            SCV_HIP(hipStreamDestroy(ctx->stream));
            ctx->own_stream = false;
        }
        ctx->stream = (hipStream_t)hip_stream;  // borrowed; NULL is the device's default stream
        return SCV_OK;
    });
}

int scv_sync(scv_ctx* ctx) {
    return guarded([&]() -> int {
        if (!ctx) return fail(SCV_ERR_ARG, "ctx is NULL");
        SCV_ENTER(ctx);
        SCV_HIP(hipStreamSynchronize(ctx->stream));
        uint32_t w = 0;
        // always read the word (4 bytes): a hot path captured into a hipGraph is replayed without passing
        // through launch_aggregate, so the host-side dirty flag says nothing about replays
        if (int rc = fetch_err(ctx, &w, true)) return rc;
        return check_err_word(ctx, w);
    });
}

int scv_set_tuning(scv_ctx* ctx, int copies, int threads, int wg_per_cu, int unroll) {
    return guarded([&]() -> int {
        if (!ctx) return fail(SCV_ERR_ARG, "ctx is NULL");
        if (copies < 0 && threads < 0 && wg_per_cu < 0 && unroll < 0) {      // all four negative: the library picks the geometry from the shape again
            ctx->copies = 16; ctx->threads = 1024; ctx->wg_per_cu = 1; ctx->unroll = 4;
            ctx->user_tuned = false;
            return SCV_OK;
        }
        const int c = copies > 0 ? copies : ctx->copies, t = threads > 0 ? threads : ctx->threads, u = unroll > 0 ? unroll : ctx->unroll;
        if (copies > 0 || threads > 0 || unroll > 0)
            if (!scv::pick_kernel(c, t, u, false, false))
                return fail(SCV_ERR_ARG, "streaming geometry copies=%d threads=%d unroll=%d is not instantiated: (copies, threads) in (4, 256) [unroll 2], "
                            "(8, 256) (8, 512) (16, 256) (16, 512) (16, 1024) [unroll 4]", c, t, u);
        ctx->copies = c; ctx->threads = t; ctx->unroll = u;
        if (wg_per_cu > 0) ctx->wg_per_cu = wg_per_cu;
        if (copies > 0 || threads > 0 || wg_per_cu > 0 || unroll > 0) ctx->user_tuned = true;   // explicit geometry wins over the auto choice
        return SCV_OK;
    });
}

int scv_set_option(scv_ctx* ctx, const char* key, int64_t value) {
    return guarded([&]() -> int {
        if (!ctx || !key) return fail(SCV_ERR_ARG, "NULL argument");
        if (!strcmp(key, "overwrite_counters")) ctx->overwrite_counters = value != 0;
        else if (!strcmp(key, "auto_geometry")) {
            // DEPRECATED alias (rounds 1-4 documented this key; round 5 replaced it by scv_set_tuning(ctx, -1, -1, -1, -1)): 1 = the library
            // picks the streaming geometry from the shape again; 0 = keep whatever scv_set_tuning set (and pin the current geometry)
            if (value) { ctx->copies = 16; ctx->threads = 1024; ctx->wg_per_cu = 1; ctx->unroll = 4; ctx->user_tuned = false; }
            else ctx->user_tuned = true;
        }
        else if (!strcmp(key, "path")) { if (value < 0 || value > 5 || value == 3) return fail(SCV_ERR_ARG, "path must be 0, 1, 2, 4 or 5"); ctx->path = (int)value; }
        else if (!strcmp(key, "sort_n_min")) { if (value < 1 || value > 65) return fail(SCV_ERR_ARG, "sort_n_min must be 1..65"); ctx->sort_n_min = (int)value; }
        else if (!strcmp(key, "sort_n_max")) { if (value < 0 || value > 64) return fail(SCV_ERR_ARG, "sort_n_max must be 0..64"); ctx->sort_n_max = (int)value; }
        else if (!strcmp(key, "reg_n_max")) { if (value < 0) return fail(SCV_ERR_ARG, "reg_n_max < 0"); ctx->reg_n_max = (int)(value > 8192 ? 8192 : value); }
        else if (!strcmp(key, "reg_shape")) { if (value < 0 || value > 9999) return fail(SCV_ERR_ARG, "reg_shape out of range"); ctx->reg_shape = (int)value; }
        else if (!strcmp(key, "fused_counters_max")) { if (value < 0) return fail(SCV_ERR_ARG, "fused_counters_max < 0"); ctx->fused_counters_max = (int)(value > (1 << 30) ? (1 << 30) : value); }
        else if (!strcmp(key, "grid")) { if (value < 0 || value > (1 << 20)) return fail(SCV_ERR_ARG, "grid out of range"); ctx->grid_override = (int)value; }
        else if (!strcmp(key, "segs")) { if (value < 0 || value > 4096) return fail(SCV_ERR_ARG, "segs out of range"); ctx->segs_override = (int)value; }
        else if (!strcmp(key, "prefix_path")) { if (value < 0 || value > 5) return fail(SCV_ERR_ARG, "prefix_path must be 0..5"); ctx->prefix_path = (int)value; }
        else if (!strcmp(key, "boot_path")) { if (value < 0 || value > 3) return fail(SCV_ERR_ARG, "boot_path must be 0..3"); ctx->boot_path = (int)value; }
        else if (!strcmp(key, "boot_spin_limit")) { if (value < 1 || value > (1 << 30)) return fail(SCV_ERR_ARG, "boot_spin_limit out of range"); ctx->boot_spin_limit = (int)value; }
        else if (!strcmp(key, "stage_mb")) { if (value < 1 || value > 65536) return fail(SCV_ERR_ARG, "stage_mb out of range"); ctx->stage_mb = (int)value; }
        else if (!strcmp(key, "host_small_kb")) { if (value < 0 || value > (1 << 20)) return fail(SCV_ERR_ARG, "host_small_kb out of range"); ctx->small_call_kb = (int)value; }
        else if (!strcmp(key, "copy_threads")) { if (value < 1 || value > 256) return fail(SCV_ERR_ARG, "copy_threads out of range"); ctx->copy_threads = (int)value; }
        else return fail(SCV_ERR_ARG, "unknown option '%s'", key);
        return SCV_OK;
    });
}

}  // extern "C" (reopened below)

namespace {

bool is_pinned_host(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

int ensure_pipe(scv_ctx* ctx, size_t bounce_bytes, size_t dslot_bytes) {
    if (!ctx->pipe) {
        ctx->pipe = new (std::nothrow) HostPipe();
        if (!ctx->pipe) return fail(SCV_ERR_ALLOC, "out of host memory");
        SCV_HIP(hipStreamCreateWithFlags(&ctx->pipe->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < HostPipe::kSlots; ++k) {
            SCV_HIP(hipEventCreateWithFlags(&ctx->pipe->landed[k], hipEventDisableTiming));
            SCV_HIP(hipEventCreateWithFlags(&ctx->pipe->consumed[k], hipEventDisableTiming));
        }
    }
    HostPipe* hp = ctx->pipe;
    int nthreads = ctx->copy_threads - 1;                      // the calling thread copies too
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && nthreads > hw - 1) nthreads = hw - 1;
    if (nthreads > 0) hp->start(nthreads, test_fault() == 1);
    if (bounce_bytes > hp->bounce_bytes) {
        for (int k = 0; k < HostPipe::kSlots; ++k) {
            if (hp->bounce[k]) { SCV_HIP(hipHostFree(hp->bounce[k])); hp->bounce[k] = nullptr; }
            SCV_HIP(hipHostMalloc(&hp->bounce[k], bounce_bytes, hipHostMallocDefault));
        }
        hp->bounce_bytes = bounce_bytes;
    }
    if (dslot_bytes > hp->dslot_bytes) {
        for (int k = 0; k < HostPipe::kSlots; ++k) {
            if (hp->dslot[k]) { SCV_HIP(hipFree(hp->dslot[k])); hp->dslot[k] = nullptr; }
            SCV_HIP(hipMalloc(&hp->dslot[k], dslot_bytes));
        }
        hp->dslot_bytes = dslot_bytes;
    }
    return SCV_OK;
}

using scv::add_copy_pieces;     // memcpy split over the pipe's worker threads (scvote_hostpool.h)

// HOST mode: the three-stage ingestion pipeline described at HostPipe.
int host_pipelined(scv_ctx* ctx, bool prefix, const int32_t* answers, const int32_t* tokens, const int32_t* n_valid,
                   const int32_t* truth, int64_t P, int32_t B, int64_t N, scv_cell* cells_out, int64_t* cell_tokens_out,
                   int64_t* tie_class_hits_out, int64_t* token_sum_out, int64_t* truth_count_sum_out) {
    auto launch = prefix ? launch_prefix : launch_dense;
    const size_t row_bytes = (prefix ? (size_t)1 : (size_t)B) * (size_t)N * sizeof(int32_t);   // votes of one problem
    const size_t row_elems = row_bytes / sizeof(int32_t);
    const size_t per_problem = row_bytes * (tokens ? 2 : 1) + sizeof(int32_t);
    const size_t total = per_problem * (size_t)P;
    // chunk: at most stage_mb, and small enough that a call of a few tens of MB still overlaps copy and DMA
    size_t target = (size_t)(ctx->stage_mb > 0 ? ctx->stage_mb : 128) << 20;
    if (total / 8 < target) target = total / 8 > ((size_t)4 << 20) ? total / 8 : ((size_t)4 << 20);
    int64_t chunk = per_problem ? (int64_t)(target / per_problem) : P;
    if (chunk < 1) chunk = 1;
    if (chunk > P) chunk = P;
    // slot layout: inputs [answers | tokens | truth], outputs [cells | cell_tokens]
    size_t off = 0;
    const size_t o_ans = off; off = align_up(off + (size_t)chunk * row_bytes, 256);
    const size_t o_tok = off; off = align_up(off + (tokens ? (size_t)chunk * row_bytes : 0), 256);
    const size_t o_truth = off; off = align_up(off + (size_t)chunk * sizeof(int32_t), 256);
    const size_t in_bytes = off;
    const size_t o_cells = off; off = align_up(off + (size_t)chunk * B * sizeof(scv_cell), 256);
    const size_t o_ctok = off; off = align_up(off + (size_t)chunk * B * sizeof(int64_t), 256);
    const size_t slot_bytes = off > 0 ? off : 256;
    if (int rc = ensure_pipe(ctx, slot_bytes, slot_bytes)) return rc;
    HostPipe* hp = ctx->pipe;
    // per-call device block: n_valid + counters
    const size_t counters_bytes = ((size_t)B * SCV_TIE_CLASSES + 2 * (size_t)B) * sizeof(int64_t);
    const size_t o_nv = 0, o_cnt = align_up((size_t)B * sizeof(int32_t), 256);
    if (int rc = ensure_stage(ctx, o_cnt + align_up(counters_bytes, 256) + 256)) return rc;
    char* sbase = static_cast<char*>(ctx->d_stage);
    int64_t* d_tie = reinterpret_cast<int64_t*>(sbase + o_cnt);
    int64_t* d_tok = d_tie + (size_t)B * SCV_TIE_CLASSES;
    int64_t* d_ts = d_tok + B;
    hipStream_t s = ctx->stream, cs = hp->copy_stream;
    SCV_HIP(hipMemsetAsync(sbase + o_cnt, 0, counters_bytes > 0 ? counters_bytes : 1, s));
    if (n_valid && B > 0) SCV_HIP(hipMemcpyAsync(sbase + o_nv, n_valid, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s));
    // buffers the caller pinned are DMA'd in place; pageable ones go through the pinned bounce slots
    const bool pin_a = is_pinned_host(answers), pin_t = !tokens || is_pinned_host(tokens);
    const int parts = ctx->copy_threads > 0 ? ctx->copy_threads : 1;

    struct InFlight { int64_t p0 = 0, pc = 0; bool used = false; } slotinfo[HostPipe::kSlots];
    auto retire = [&](int k) -> int {                      // chunk in slot k is done: hand its cell table to the caller
        if (!slotinfo[k].used) return SCV_OK;
        SCV_HIP(hipEventSynchronize(hp->consumed[k]));
        const char* bb = static_cast<const char*>(hp->bounce[k]);
        const int64_t p0 = slotinfo[k].p0, pc = slotinfo[k].pc;
        if (cells_out && B > 0) memcpy(cells_out + (size_t)p0 * B, bb + o_cells, (size_t)pc * B * sizeof(scv_cell));
        if (cell_tokens_out && B > 0) memcpy(cell_tokens_out + (size_t)p0 * B, bb + o_ctok, (size_t)pc * B * sizeof(int64_t));
        slotinfo[k].used = false;
        return SCV_OK;
    };

    int64_t idx = 0;
    for (int64_t p0 = 0; p0 < P; p0 += chunk, ++idx) {
        const int k = (int)(idx % HostPipe::kSlots);
        const int64_t pc = (P - p0 < chunk) ? (P - p0) : chunk;
        if (int rc = retire(k)) return rc;                 // slot k (bounce + HBM) is free again
        char* bb = static_cast<char*>(hp->bounce[k]);
        char* db = static_cast<char*>(hp->dslot[k]);
        // stage 1: caller memory -> pinned bounce slot (worker threads); overlaps the DMA of the previous chunk
        std::vector<std::function<void()>> pieces;
        if (const int tf = test_fault(); tf == 2 || tf == 3) { if (tf == 2) throw std::bad_alloc(); throw std::runtime_error("test hook: SCV_TEST_FAULT=throw"); }
#ifdef SCV_TEST_HOOKS
        if (test_fault() == 4) {                               // (test builds only) unsynchronised increments from the copy workers
            static volatile long racy = 0;
            for (int r = 0; r < 8; ++r) pieces.emplace_back([] { for (int k = 0; k < 20000; ++k) racy = racy + 1; });
        }
#endif
        if (!pin_a) add_copy_pieces(pieces, bb + o_ans, answers + (size_t)p0 * row_elems, (size_t)pc * row_bytes, parts);
        if (tokens && !pin_t) add_copy_pieces(pieces, bb + o_tok, tokens + (size_t)p0 * row_elems, (size_t)pc * row_bytes, parts);
        memcpy(bb + o_truth, truth + p0, (size_t)pc * sizeof(int32_t));
        hp->run(std::move(pieces));
        // stage 2: DMA to the HBM slot on the copy stream
        const void* src_a = pin_a ? static_cast<const void*>(answers + (size_t)p0 * row_elems) : static_cast<const void*>(bb + o_ans);
        if (row_bytes) SCV_HIP(hipMemcpyAsync(db + o_ans, src_a, (size_t)pc * row_bytes, hipMemcpyHostToDevice, cs));
        if (tokens && row_bytes) {
            const void* src_t = pin_t ? static_cast<const void*>(tokens + (size_t)p0 * row_elems) : static_cast<const void*>(bb + o_tok);
            SCV_HIP(hipMemcpyAsync(db + o_tok, src_t, (size_t)pc * row_bytes, hipMemcpyHostToDevice, cs));
        }
        SCV_HIP(hipMemcpyAsync(db + o_truth, bb + o_truth, (size_t)pc * sizeof(int32_t), hipMemcpyHostToDevice, cs));
        SCV_HIP(hipEventRecord(hp->landed[k], cs));
        // stage 3: hot path on the compute stream, cell table back into the pinned slot
        SCV_HIP(hipStreamWaitEvent(s, hp->landed[k], 0));
        if (int rc = launch(ctx, reinterpret_cast<const int32_t*>(db + o_ans),
                            tokens ? reinterpret_cast<const int32_t*>(db + o_tok) : nullptr,
                            n_valid ? reinterpret_cast<const int32_t*>(sbase + o_nv) : nullptr,
                            reinterpret_cast<const int32_t*>(db + o_truth), pc, B, N,
                            reinterpret_cast<scv_cell*>(db + o_cells), reinterpret_cast<int64_t*>(db + o_ctok), d_tie, d_tok, d_ts))
            return rc;
        if (cells_out && B > 0) SCV_HIP(hipMemcpyAsync(bb + o_cells, db + o_cells, (size_t)pc * B * sizeof(scv_cell), hipMemcpyDeviceToHost, s));
        if (cell_tokens_out && B > 0) SCV_HIP(hipMemcpyAsync(bb + o_ctok, db + o_ctok, (size_t)pc * B * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        SCV_HIP(hipEventRecord(hp->consumed[k], s));
        // the copy stream may not overwrite this HBM slot before the kernel has consumed it (next use: chunk idx + 2,
        // whose retire(k) synchronises the host on consumed[k] before anything is enqueued)
        slotinfo[k].p0 = p0; slotinfo[k].pc = pc; slotinfo[k].used = true;
    }
    for (int k = 0; k < HostPipe::kSlots; ++k)
        if (int rc = retire(k)) return rc;
    if (B > 0) {
        if (tie_class_hits_out) SCV_HIP(hipMemcpyAsync(tie_class_hits_out, d_tie, (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        if (token_sum_out) SCV_HIP(hipMemcpyAsync(token_sum_out, d_tok, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        if (truth_count_sum_out) SCV_HIP(hipMemcpyAsync(truth_count_sum_out, d_ts, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    }
    SCV_HIP(hipStreamSynchronize(s));
    uint32_t w = 0;
    if (int rc = fetch_err(ctx, &w)) return rc;
    return check_err_word(ctx, w);
}

// HOST mode, small calls.  Every call the reference itself makes is tiny (P = 30 problems, N <= 128 samples, o1.py:277,302: a 15 KB
// tensor and a 10 us kernel), and the three-stage pipeline above -- built for GB-sized inputs -- cost it 0.25-0.4 ms of thread
// hand-offs, events and per-chunk copies.  Here the whole call is ONE pinned block and ONE HBM block owned by the ctx:
//     [ answers | tokens | truth | n_valid | counters = 0 | error word = 0 | cells | cell_tokens ]
//       `---------------- one H2D -------------------------------------'
//                                            `------------------- one D2H ------------------------'
// host memcpy in, one hipMemcpyAsync, the kernel, one hipMemcpyAsync back, one stream sync, host memcpy out.  No worker threads,
// no bounce slots, no per-call allocation; the kernels' error flag points into the block for the duration of the call, so the word
// arrives with the results and is zero again on the next call without a memset.
int host_small(scv_ctx* ctx, bool prefix, const int32_t* answers, const int32_t* tokens, const int32_t* n_valid,
               const int32_t* truth, int64_t P, int32_t B, int64_t N, scv_cell* cells_out, int64_t* cell_tokens_out,
               int64_t* tie_class_hits_out, int64_t* token_sum_out, int64_t* truth_count_sum_out, bool* taken) {
    *taken = false;
    if (ctx->small_call_kb <= 0 || P <= 0 || B <= 0) return SCV_OK;
    const size_t votes_bytes = (size_t)P * (prefix ? (size_t)1 : (size_t)B) * (size_t)N * sizeof(int32_t);
    const size_t counters_bytes = ((size_t)B * SCV_TIE_CLASSES + 2 * (size_t)B) * sizeof(int64_t);
    size_t off = 0;
    const size_t o_ans = off; off = align_up(off + votes_bytes, 256);
    const size_t o_tok = off; off = align_up(off + (tokens ? votes_bytes : 0), 256);
    const size_t o_truth = off; off = align_up(off + (size_t)P * sizeof(int32_t), 256);
    const size_t o_nv = off; off = align_up(off + (size_t)B * sizeof(int32_t), 256);
    const size_t o_cnt = off; off = align_up(off + counters_bytes, 256);
    const size_t o_err = off; off += 256;
    const size_t h2d_bytes = off;
    const size_t o_cells = off; off = align_up(off + (size_t)P * B * sizeof(scv_cell), 256);
    const size_t o_ctok = off; off = align_up(off + (tokens ? (size_t)P * B * sizeof(int64_t) : 0), 256);
    const size_t total = off;
    if (total > (size_t)ctx->small_call_kb << 10) return SCV_OK;
    if (total > ctx->small_bytes) {
        const size_t want = ((size_t)ctx->small_call_kb << 10) > total ? ((size_t)ctx->small_call_kb << 10) : total;
        if (ctx->small_h) { SCV_HIP(hipHostFree(ctx->small_h)); ctx->small_h = nullptr; }
        if (ctx->small_d) { SCV_HIP(hipFree(ctx->small_d)); ctx->small_d = nullptr; }
        ctx->small_bytes = 0;
        SCV_HIP(hipHostMalloc(&ctx->small_h, want, hipHostMallocDefault));
        SCV_HIP(hipMalloc(&ctx->small_d, want));
        ctx->small_bytes = want;
    }
    *taken = true;
    ctx->stat_small_calls += 1;
    char* hb = static_cast<char*>(ctx->small_h);
    char* db = static_cast<char*>(ctx->small_d);
    if (votes_bytes) memcpy(hb + o_ans, answers, votes_bytes);
    if (tokens && votes_bytes) memcpy(hb + o_tok, tokens, votes_bytes);
    memcpy(hb + o_truth, truth, (size_t)P * sizeof(int32_t));
    if (n_valid) memcpy(hb + o_nv, n_valid, (size_t)B * sizeof(int32_t));
    memset(hb + o_cnt, 0, o_err + 256 - o_cnt);                       // counters + error word start at zero on the device too
    hipStream_t s = ctx->stream;
    SCV_HIP(hipMemcpyAsync(db, hb, h2d_bytes, hipMemcpyHostToDevice, s));
    int64_t* d_tie = reinterpret_cast<int64_t*>(db + o_cnt);
    int64_t* d_tok = d_tie + (size_t)B * SCV_TIE_CLASSES;
    int64_t* d_ts = d_tok + B;
    // the kernels report into the block's own error word for this call (restored on every path)
    struct ErrSwap { scv_ctx* c; uint32_t* keep; bool dirty; ~ErrSwap() { c->d_err = keep; c->err_dirty = dirty; } } swap{ctx, ctx->d_err, ctx->err_dirty};
    ctx->d_err = reinterpret_cast<uint32_t*>(db + o_err);
    auto launch = prefix ? launch_prefix : launch_dense;
    if (int rc = launch(ctx, reinterpret_cast<const int32_t*>(db + o_ans), tokens ? reinterpret_cast<const int32_t*>(db + o_tok) : nullptr,
                        n_valid ? reinterpret_cast<const int32_t*>(db + o_nv) : nullptr, reinterpret_cast<const int32_t*>(db + o_truth), P, B, N,
                        reinterpret_cast<scv_cell*>(db + o_cells), tokens ? reinterpret_cast<int64_t*>(db + o_ctok) : nullptr, d_tie, d_tok, d_ts)) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    const bool want_cells = cells_out || cell_tokens_out;
    SCV_HIP(hipMemcpyAsync(hb + o_cnt, db + o_cnt, (want_cells ? total : o_cells) - o_cnt, hipMemcpyDeviceToHost, s));
    SCV_HIP(hipStreamSynchronize(s));
    uint32_t w = 0;
    memcpy(&w, hb + o_err, sizeof w);
    if (tie_class_hits_out) memcpy(tie_class_hits_out, hb + o_cnt, (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t));
    if (token_sum_out) memcpy(token_sum_out, hb + o_cnt + (size_t)B * SCV_TIE_CLASSES * sizeof(int64_t), (size_t)B * sizeof(int64_t));
    if (truth_count_sum_out) memcpy(truth_count_sum_out, hb + o_cnt + ((size_t)B * SCV_TIE_CLASSES + B) * sizeof(int64_t), (size_t)B * sizeof(int64_t));
    if (cells_out) memcpy(cells_out, hb + o_cells, (size_t)P * B * sizeof(scv_cell));
    if (cell_tokens_out && tokens) memcpy(cell_tokens_out, hb + o_ctok, (size_t)P * B * sizeof(int64_t));
    return check_err_word(ctx, w);
}

// Shared body of scv_aggregate_i32 (dense: rows of B*N votes per problem) and
// scv_aggregate_prefix_i32 (prefix: one row of N votes per problem).
int aggregate_common(scv_ctx* ctx, bool prefix, const int32_t* answers, const int32_t* tokens, const int32_t* n_valid,
                     const int32_t* truth, int64_t P, int32_t B, int64_t N, int mem_kind, scv_cell* cells_out,
                     int64_t* cell_tokens_out, int64_t* tie_class_hits_out, int64_t* token_sum_out,
                     int64_t* truth_count_sum_out) {
    if (!ctx) return fail(SCV_ERR_ARG, "ctx is NULL");
    if (P < 0 || B < 0 || N < 0) return fail(SCV_ERR_ARG, "negative shape P=%lld B=%d N=%lld", (long long)P, B, (long long)N);
    if (N > 0x7fffffffll) return fail(SCV_ERR_ARG, "N=%lld exceeds 2^31-1 (cell counts are u32)", (long long)N);
    if (P > 0 && B > 0 && !truth) return fail(SCV_ERR_ARG, "truth is NULL");
    if (P > 0 && B > 0 && N > 0 && !answers) return fail(SCV_ERR_ARG, "answers is NULL");
    if (prefix && B > 0 && !n_valid) return fail(SCV_ERR_ARG, "prefix mode needs n_valid");
    if (prefix && B > scv::kMaxSortedB) return fail(SCV_ERR_ARG, "prefix mode supports at most %d budgets", scv::kMaxSortedB);
    if (mem_kind != SCV_MEM_HOST && mem_kind != SCV_MEM_DEVICE) return fail(SCV_ERR_ARG, "bad mem_kind %d", mem_kind);
    if ((ctx->flags & SCV_FLAG_PACKED_CELLS) && cells_out) {
        // 4-byte records: DEVICE-mode scv_aggregate_i32 over cells of up to 127 votes (every count fits 7 bits) on the kernels that serve such cells
        if (prefix || mem_kind != SCV_MEM_DEVICE || N > 127 || ctx->path == 1 || ctx->path == 2 || ctx->fused_counters_max == 0 || ctx->boot_req)
            return fail(SCV_ERR_ARG, "SCV_FLAG_PACKED_CELLS: cells_out is uint32 [P, B] only for scv_aggregate_i32 on DEVICE memory with N <= 127 "
                                     "(auto dispatch; no prefix budgets, no bootstrap in the call); got prefix=%d mem_kind=%d N=%lld", (int)prefix, mem_kind, (long long)N);
    }
    SCV_ENTER(ctx);
    auto launch = prefix ? launch_prefix : launch_dense;

    if (mem_kind == SCV_MEM_DEVICE)
        return launch(ctx, answers, tokens, n_valid, truth, P, B, N, cells_out, cell_tokens_out,
                      tie_class_hits_out, token_sum_out, truth_count_sum_out);

    // HOST mode accumulates its own zeroed counters over the chunks: the DEVICE-mode overwrite option must not apply
    struct Restore { scv_ctx* c; int v; ~Restore() { c->overwrite_counters = v; c->nv_host = nullptr; } } restore{ctx, ctx->overwrite_counters};
    ctx->overwrite_counters = 0;
    ctx->nv_host = prefix ? n_valid : nullptr;
    bool small = false;
    if (int rc = host_small(ctx, prefix, answers, tokens, n_valid, truth, P, B, N, cells_out, cell_tokens_out,
                            tie_class_hits_out, token_sum_out, truth_count_sum_out, &small)) return rc;
    if (small) return SCV_OK;
    ctx->stat_pipelined_calls += 1;
    return host_pipelined(ctx, prefix, answers, tokens, n_valid, truth, P, B, N, cells_out, cell_tokens_out,
                          tie_class_hits_out, token_sum_out, truth_count_sum_out);
}

}  // namespace

extern "C" {

int scv_aggregate_i32(scv_ctx* ctx, const int32_t* answers, const int32_t* tokens, const int32_t* n_valid,
                      const int32_t* truth, int64_t P, int32_t B, int64_t N, int mem_kind, scv_cell* cells_out,
                      int64_t* cell_tokens_out, int64_t* tie_class_hits_out, int64_t* token_sum_out,
                      int64_t* truth_count_sum_out) {
    return guarded([&]() -> int {
        return aggregate_common(ctx, false, answers, tokens, n_valid, truth, P, B, N, mem_kind, cells_out, cell_tokens_out,
                                tie_class_hits_out, token_sum_out, truth_count_sum_out);
    });
}

int scv_aggregate_prefix_i32(scv_ctx* ctx, const int32_t* pool, const int32_t* tokens, const int32_t* n_valid,
                             const int32_t* truth, int64_t P, int32_t B, int64_t N, int mem_kind, scv_cell* cells_out,
                             int64_t* cell_tokens_out, int64_t* tie_class_hits_out, int64_t* token_sum_out,
                             int64_t* truth_count_sum_out) {
    return guarded([&]() -> int {
        return aggregate_common(ctx, true, pool, tokens, n_valid, truth, P, B, N, mem_kind, cells_out, cell_tokens_out,
                                tie_class_hits_out, token_sum_out, truth_count_sum_out);
    });
}

int scv_bootstrap(scv_ctx* ctx, const scv_cell* cells, int64_t P, int32_t B, int32_t r_begin, int32_t r_end,
                  uint64_t seed, int32_t M, int mem_kind, int64_t* counts_out) {
    return guarded([&]() -> int {
        if (!ctx) return fail(SCV_ERR_ARG, "ctx is NULL");
        if (!cells || !counts_out) return fail(SCV_ERR_ARG, "bootstrap: NULL pointer");
        if (P <= 0 || P > 0xFFFFFFFFll || B <= 0 || M <= 0 || r_end < r_begin || r_begin < 0)
            return fail(SCV_ERR_ARG, "bootstrap: bad shape P=%lld B=%d M=%d r=[%d,%d)", (long long)P, B, M, r_begin, r_end);
        const size_t lds = (size_t)B * M * sizeof(uint32_t);
        if (lds > 64 * 1024) return fail(SCV_ERR_ARG, "bootstrap: B*M=%lld counters exceed 64 KiB of LDS", (long long)B * M);
        if (mem_kind != SCV_MEM_HOST && mem_kind != SCV_MEM_DEVICE) return fail(SCV_ERR_ARG, "bad mem_kind %d", mem_kind);
        SCV_ENTER(ctx);
        const int32_t R = r_end - r_begin;
        if (R == 0) return SCV_OK;
        hipStream_t s = ctx->stream;
        const size_t out_bytes = (size_t)R * B * M * sizeof(int64_t);
        // LDS-resident kernel when the 2-byte code table + counters fit (P * B up to ~70 k cells); otherwise the
        // global-gather kernel.  "boot_path" = 3 forces the latter (parity tests).
        const size_t lds_fast = (((size_t)B * M + 3) & ~(size_t)3) * sizeof(uint32_t) + (((size_t)P * B + 7) & ~(size_t)7) * sizeof(uint16_t);
        const bool fast = ctx->boot_path != 3 && lds_fast <= (size_t)144 * 1024;
        auto launch_boot = [&](const scv_cell* d_cells, unsigned long long* d_out) -> int {
            if (fast) {
                int64_t grid = (int64_t)ctx->num_cus;                          // one 1024-thread workgroup per CU, R / grid resamples each
                if (grid > R) grid = R;
                SCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scv::scv_bootstrap_lds_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast));
                hipLaunchKernelGGL(scv::scv_bootstrap_lds_k, dim3((unsigned)grid), dim3(1024), lds_fast, s, d_cells, P, B, r_begin, r_end, seed, M, d_out, ctx->d_err);
            } else {
                hipLaunchKernelGGL(scv::scv_bootstrap_k, dim3((unsigned)R), dim3(256), lds, s, d_cells, P, B, r_begin, seed, M, d_out, ctx->d_err);
            }
            SCV_HIP(hipGetLastError());
            ctx->err_dirty = true;
            return SCV_OK;
        };
        if (mem_kind == SCV_MEM_DEVICE) return launch_boot(cells, reinterpret_cast<unsigned long long*>(counts_out));
        const size_t cells_bytes = (size_t)P * B * sizeof(scv_cell);
        const size_t o_out = align_up(cells_bytes, 256);
        if (int rc = ensure_stage(ctx, o_out + out_bytes)) return rc;
        char* base = static_cast<char*>(ctx->d_stage);
        SCV_HIP(hipMemcpyAsync(base, cells, cells_bytes, hipMemcpyHostToDevice, s));
        if (int rc = launch_boot(reinterpret_cast<const scv_cell*>(base), reinterpret_cast<unsigned long long*>(base + o_out))) return rc;
        SCV_HIP(hipMemcpyAsync(counts_out, base + o_out, out_bytes, hipMemcpyDeviceToHost, s));
        SCV_HIP(hipStreamSynchronize(s));
        uint32_t w = 0;
        if (int rc = fetch_err(ctx, &w)) return rc;
        return check_err_word(ctx, w);
    });
}

int scv_aggregate_bootstrap_i32(scv_ctx* ctx, const int32_t* answers, const int32_t* tokens, const int32_t* n_valid,
                                const int32_t* truth, int64_t P, int32_t B, int64_t N, scv_cell* cells_out,
                                int64_t* cell_tokens_out, int64_t* tie_class_hits_out, int64_t* token_sum_out,
                                int64_t* truth_count_sum_out, int32_t r_begin, int32_t r_end, uint64_t seed, int32_t M,
                                int64_t* counts_out) {
    return guarded([&]() -> int {
        if (!ctx) return fail(SCV_ERR_ARG, "ctx is NULL");
        if (!cells_out || !counts_out) return fail(SCV_ERR_ARG, "aggregate_bootstrap: cells_out and counts_out are required");
        if (ctx->flags & SCV_FLAG_PACKED_CELLS) return fail(SCV_ERR_ARG, "SCV_FLAG_PACKED_CELLS: vote + bootstrap in one call reads 16-byte records (use a ctx without the flag)");
        if (P <= 0 || P > 0xFFFFFFFFll || B <= 0 || M <= 0 || r_end < r_begin || r_begin < 0)
            return fail(SCV_ERR_ARG, "aggregate_bootstrap: bad shape P=%lld B=%d M=%d r=[%d,%d)", (long long)P, B, M, r_begin, r_end);
        if ((size_t)B * M * sizeof(uint32_t) > 64 * 1024) return fail(SCV_ERR_ARG, "bootstrap: B*M=%lld counters exceed 64 KiB of LDS", (long long)B * M);
        scv_ctx::BootReq rq{r_begin, r_end, M, seed, counts_out, false};
        if (r_end > r_begin) ctx->boot_req = &rq;
        const int rc = scv_aggregate_i32(ctx, answers, tokens, n_valid, truth, P, B, N, SCV_MEM_DEVICE, cells_out, cell_tokens_out,
                                         tie_class_hits_out, token_sum_out, truth_count_sum_out);
        ctx->boot_req = nullptr;
        if (rc != SCV_OK || rq.fused || r_end == r_begin) return rc;
        // shape or occupancy did not allow the fused form: the bootstrap is queued behind the vote on the same stream
        ctx->stat_boot_separate += 1;
        return scv_bootstrap(ctx, cells_out, P, B, r_begin, r_end, seed, M, SCV_MEM_DEVICE, counts_out);
    });
}

int scv_synth_fill_i32(scv_ctx* ctx, int32_t* answers, int32_t* tokens, int32_t* truth, int64_t P, int32_t B,
                       int64_t N, int64_t p_offset, uint64_t seed, int dist) {
    return guarded([&]() -> int {
        if (!ctx) return fail(SCV_ERR_ARG, "ctx is NULL");
        if (P < 0 || B < 0 || N < 0 || p_offset < 0) return fail(SCV_ERR_ARG, "synth_fill: negative shape");
        if (dist < SCV_DIST_UNIFORM || dist > SCV_DIST_DEGENERATE_WRONG) return fail(SCV_ERR_ARG, "synth_fill: unknown dist %d", dist);
        SCV_ENTER(ctx);
        if (P == 0) return SCV_OK;
        int64_t grid = P * (int64_t)B;
        if (B == 0) grid = (P + 255) / 256;
        const int64_t cap = (int64_t)ctx->num_cus * 16;
        if (grid > cap) grid = cap;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(scv::scv_synth_fill_k, dim3((unsigned)grid), dim3(256), 0, ctx->stream, answers, tokens, truth, P,
                           B, N, p_offset, seed, dist);
        SCV_HIP(hipGetLastError());
        return SCV_OK;
    });
}

int scv_export_error_word(scv_ctx* ctx, int64_t* dst_device) {
    return guarded([&]() -> int {
        if (!ctx || !dst_device) return fail(SCV_ERR_ARG, "NULL argument");
        SCV_ENTER(ctx);
        // the word as scv_sync would JUDGE it: under SCV_FLAG_CLAMP_TO_INVALID_BIN an out-of-domain vote is not an error (bit 0 dropped)
        const uint32_t mask = (ctx->flags & SCV_FLAG_CLAMP_TO_INVALID_BIN) ? ~1u : ~0u;
        hipLaunchKernelGGL(scv::scv_export_err_k, dim3(1), dim3(1), 0, ctx->stream, ctx->d_err, mask, reinterpret_cast<long long*>(dst_device));
        SCV_HIP(hipGetLastError());
        return SCV_OK;
    });
}

int scv_last_kernel_ns(scv_ctx* ctx, uint64_t* ns_out) {
    return guarded([&]() -> int {
        if (!ctx || !ns_out) return fail(SCV_ERR_ARG, "NULL argument");
        if (!(ctx->flags & SCV_FLAG_TIMING) || ctx->events_used == 0)
            return fail(SCV_ERR_NOT_TIMED, "no timed launch (create the ctx with SCV_FLAG_TIMING)");
        SCV_ENTER(ctx);
        EventPair& ev = ctx->events[ctx->events_used - 1];
        SCV_HIP(hipEventSynchronize(ev.b));
        float ms = 0.f;
        SCV_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
        *ns_out = (uint64_t)((double)ms * 1e6);
        return SCV_OK;
    });
}

int scv_drain_kernel_ns(scv_ctx* ctx, uint64_t* total_ns_out, uint64_t* launches_out) {
    return guarded([&]() -> int {
        if (!ctx || !total_ns_out || !launches_out) return fail(SCV_ERR_ARG, "NULL argument");
        if (!(ctx->flags & SCV_FLAG_TIMING)) return fail(SCV_ERR_NOT_TIMED, "ctx was created without SCV_FLAG_TIMING");
        SCV_ENTER(ctx);
        double total = 0;
        for (size_t i = 0; i < ctx->events_used; ++i) {
            SCV_HIP(hipEventSynchronize(ctx->events[i].b));
            float ms = 0.f;
            SCV_HIP(hipEventElapsedTime(&ms, ctx->events[i].a, ctx->events[i].b));
            total += (double)ms * 1e6;
        }
        *total_ns_out = (uint64_t)total;
        *launches_out = ctx->events_used;
        ctx->events_used = 0;
        return SCV_OK;
    });
}

int scv_host_alloc(void** out, size_t bytes) {
    return guarded([&]() -> int {
        if (!out) return fail(SCV_ERR_ARG, "scv_host_alloc: out is NULL");
        *out = nullptr;
        if (bytes == 0) return SCV_OK;
        SCV_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
        return SCV_OK;
    });
}

int scv_host_free(void* p) {
    return guarded([&]() -> int {
        if (!p) return SCV_OK;
        SCV_HIP(hipHostFree(p));
        return SCV_OK;
    });
}

int scv_get_stat(scv_ctx* ctx, const char* key, int64_t* out) {
    return guarded([&]() -> int {
        if (!ctx || !key || !out) return fail(SCV_ERR_ARG, "NULL argument");
        if (!strcmp(key, "boot_fused")) *out = ctx->stat_boot_fused;
        else if (!strcmp(key, "boot_separate")) *out = ctx->stat_boot_separate;
        else if (!strcmp(key, "boot_recovered")) *out = ctx->stat_boot_recovered;
        else if (!strcmp(key, "boot_cooperative")) *out = ctx->stat_boot_cooperative;
        else if (!strcmp(key, "overwrite_fused")) *out = ctx->stat_overwrite_fused;
        else if (!strcmp(key, "lds_counters")) *out = ctx->stat_lds_counters;
        else if (!strcmp(key, "prefix_cells")) *out = ctx->stat_prefix_cells;
        else if (!strcmp(key, "prefix_lane")) *out = ctx->stat_prefix_lane;
    else if (!strcmp(key, "prefix_pool")) *out = ctx->stat_prefix_pool;
    else if (!strcmp(key, "prefix_sort")) *out = ctx->stat_prefix_sort;
    else if (!strcmp(key, "prefix_tokens")) *out = ctx->stat_prefix_tokens;
    else if (!strcmp(key, "one_vote")) *out = ctx->stat_one_vote;
        else if (!strcmp(key, "sort_cells")) *out = ctx->stat_sort_cells;
        else if (!strcmp(key, "few_votes")) *out = ctx->stat_few_votes;
        else if (!strcmp(key, "host_small_calls")) *out = ctx->stat_small_calls;
        else if (!strcmp(key, "host_pipelined_calls")) *out = ctx->stat_pipelined_calls;
        else if (!strcmp(key, "host_thread_start_failures")) *out = ctx->pipe ? ctx->pipe->pool.start_failures : 0;
        else return fail(SCV_ERR_ARG, "unknown stat '%s'", key);
        return SCV_OK;
    });
}

int scv_device_info(scv_ctx* ctx, int64_t info_out[4]) {
    return guarded([&]() -> int {
        if (!ctx || !info_out) return fail(SCV_ERR_ARG, "NULL argument");
        info_out[0] = ctx->num_cus;
        info_out[1] = ctx->lds_max;
        info_out[2] = ctx->clock_khz;
        info_out[3] = ctx->hbm_bytes;
        return SCV_OK;
    });
}

}  // extern "C"

// ---- what csrc/scvote_comm.hip needs of the opaque context -----------------------------------------------------------
namespace scv {
hipStream_t ctx_stream(scv_ctx* ctx) { return ctx->stream; }
int ctx_device(scv_ctx* ctx) { return ctx->device; }
int comm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace scv

// ---- kernel tables: the per-translation-unit tables behind one switch each (scvote_dispatch.h) ----------------------
namespace scv {
KernelFn pick_kernel(int copies, int t, int u, bool tok, bool xtra) {
    switch (copies) {
    case 4: return pick_stream_c4(t, u, tok, xtra);
    case 8: return pick_stream_c8(t, u, tok, xtra);
    case 16: return pick_stream_c16(t, u, tok, xtra);
    default: return nullptr;
    }
}
RegKernel pick_reg_kernel(int g, int v, bool tok, bool vec) {
    if (g == 8) return pick_reg_g8(v, tok, vec);
    if (g == 16) return pick_reg_g16(v, tok, vec);
    if (g == 32) return pick_reg_g32(v, tok, vec);
    return pick_reg_g64(v, tok, vec);
}
}  // namespace scv
