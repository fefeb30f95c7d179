// scvote_comm.hip -- the exchange step of the path behind the C ABI (include/scvote.h: scv_comm_*, scv_allreduce_counters).
//
// The reference is ONE process (o1.py:312-315) that sums its per-problem scores in a Python loop (o1.py:236-245).  When the
// problems are sharded over the GPUs of a node, that sum is one all-reduce of the packed int64 per-budget counters (65.7 KB at
// B = 8: latency-bound).  A single-process caller (ctypes, C) gets it here without torch.distributed:
//
//   SCV_COMM_PEER (default)  one-shot all-reduce over xGMI peer access, pure HIP: every rank's kernel reads the other ranks'
//                            buffers directly (hipDeviceEnablePeerAccess), sums them into a staging buffer of its own, and copies
//                            the sum back when every rank has finished reading.  Ordering between the devices' streams is by
//                            events (hipStreamWaitEvent across devices); the host never blocks.  G buffers of 65.7 KB cross each
//                            link once: with 7 links x ~153 GB/s per GPU the exchange is two kernel launches of latency.
//   SCV_COMM_RCCL            ncclCommInitAll + grouped ncclAllReduce(ncclInt64, ncclSum) on the ranks' streams.  librccl is
//                            resolved at run time (the copy the process already holds -- torch's -- or /opt/rocm's): the library
//                            has no link-time dependency on it.
//
// Integer sums: the result is independent of the order of the ranks => bit-exact at any G.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <vector>

#include "../../include/scvote.h"

namespace scv {
// accessors of the opaque context (csrc/scvote.hip)
hipStream_t ctx_stream(scv_ctx* ctx);
int ctx_device(scv_ctx* ctx);
int comm_fail(int code, const char* fmt, ...);   // sets scv_last_error's thread-local message

constexpr int kMaxRanks = 16;
struct PeerPtrs { const long long* p[kMaxRanks]; };

// How a kernel reads another device's (coarse-grained hipMalloc) buffer.  NT = __builtin_nontemporal_load (global_load ... nt: no
// allocation in this device's caches, so no stale line can be kept between two exchanges); the ordinary load is the alternative the
// create-time self-test also tries on distinct devices.  Which of the two the communicator uses is decided by that test
// (scv_comm_get_stat "peer_loads"), not assumed.
template <bool NT>
__device__ __forceinline__ long long peer_load(const long long* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *reinterpret_cast<const volatile long long*>(p);
}

// out[i] = sum over the ranks of in_r[i]; the other ranks' buffers are read through peer access (xGMI)
template <bool NT>
__global__ __launch_bounds__(256) void scv_sum_peers_k(PeerPtrs in, int n, long long* out, int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        long long s = 0;
        for (int r = 0; r < n; ++r) s += peer_load<NT>(in.p[r] + i);
        out[i] = s;
    }
}

// ---- self-test of a new communicator (scv_comm_create) ---------------------------------------------------------------------
// pattern of rank r, round k, word i: every rank / round / word differs, sums over ranks are closed-form
__device__ __host__ inline long long comm_pattern(int r, int k, int64_t i) {
    return (long long)(r + 1) * 0x100000001ll + (long long)i * (2 * r + 1) + (long long)k * 0x10001ll * (r + 3);
}
__global__ __launch_bounds__(256) void scv_comm_fill_k(long long* buf, int r, int k, int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) buf[i] = comm_pattern(r, k, i);
}
// res[0] += words of `buf` that differ from `rank_lo..rank_hi`'s summed patterns; res[1] = min(first differing word).
// `skew` is 0 except under the test hook SCV_TEST_FAULT=peer, which makes every expectation wrong (a self-test that must fail).
template <bool NT>
__global__ __launch_bounds__(256) void scv_comm_verify_k(const long long* buf, int rank_lo, int rank_hi, int k, int64_t count, unsigned long long* res, long long skew) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        long long want = skew;
        for (int r = rank_lo; r < rank_hi; ++r) want += comm_pattern(r, k, i);
        if (peer_load<NT>(buf + i) != want) { atomicAdd(&res[0], 1ull); atomicMin(&res[1], (unsigned long long)i); }
    }
}
}  // namespace scv

// ---- RCCL, resolved at run time -------------------------------------------------------------------------------------
namespace {
typedef void* ncclComm_t;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // the copy the process already has
        if (!lib) for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllReduce && Broadcast && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;
constexpr int kNcclInt8 = 0, kNcclInt64 = 4, kNcclSum = 0;     // rccl.h: ncclDataType_t / ncclRedOp_t

struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

struct scv_comm {
    int n = 0;
    uint32_t flags = 0;
    std::vector<int> devices;
    std::vector<scv_ctx*> ctx;
    std::vector<hipEvent_t> ready, done;
    std::vector<void*> tmp;
    size_t tmp_bytes = 0;
    std::vector<ncclComm_t> nccl;
    int64_t stat_selftest_words = 0;     // words verified per rank by the create-time self-test (0: one rank, nothing to test)
    // SCV_COMM_PEER: how scv_sum_peers_k reads the peers' buffers -- 0 nontemporal loads (the design), 1 ordinary loads (used only when
    // the self-test saw nontemporal reads fail and ordinary ones pass), -1 no peer reads in this communicator (RCCL / one rank)
    int peer_loads = -1;
    int selftest_nt_ok = -1, selftest_plain_ok = -1;     // pairwise peer reads of the self-test: 1 all correct, 0 some wrong, -1 not run
};

namespace {
constexpr size_t kTmpBytesAtCreate = 1 << 20;   // staging buffer of the one-shot all-reduce: 131072 int64 (B <= 127 budgets' counters) without
                                                // ever allocating on the launch path (legal under hipGraph capture)
int comm_selftest(scv_comm* c);
}

// No C++ exception crosses the C ABI (std::vector growth in here can throw std::bad_alloc): every extern "C" body runs inside guarded().
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return scv::comm_fail(SCV_ERR_ALLOC, "out of host memory (std::bad_alloc inside the library)");
    } catch (const std::exception& e) {
        return scv::comm_fail(SCV_ERR_ARG, "internal error: %s", e.what());
    } catch (...) {
        return scv::comm_fail(SCV_ERR_ARG, "internal error: unknown C++ exception");
    }
}

// TEST HOOK (only in builds with -DSCV_TEST_HOOKS: csrc/libscvote_hooks.so): SCV_TEST_FAULT=peer makes the self-test of a SCV_COMM_PEER communicator fail (every expectation skewed), and makes it run
// for one rank too -- so that a 1-GPU box can exercise "self-test failed -> RCCL" (MultiDeviceEngine, tests/test_gpu_parity.py).
static bool peer_fault_for_test() {
#ifdef SCV_TEST_HOOKS
    const char* f = getenv("SCV_TEST_FAULT");
    return f && !strcmp(f, "peer");
#else
    return false;
#endif
}

#define COMM_HIP(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return scv::comm_fail(e_ == hipErrorOutOfMemory ? SCV_ERR_ALLOC : -(int)e_, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int scv_comm_destroy(scv_comm* c) {
    return guarded([&]() -> int {
        if (!c) return SCV_OK;
        DeviceScope scope;
        for (int r = 0; r < (int)c->ctx.size(); ++r) {
            (void)hipSetDevice(c->devices[r]);
            if (r < (int)c->nccl.size() && c->nccl[r] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl[r]);
            if (r < (int)c->ready.size() && c->ready[r]) (void)hipEventDestroy(c->ready[r]);
            if (r < (int)c->done.size() && c->done[r]) (void)hipEventDestroy(c->done[r]);
            if (r < (int)c->tmp.size() && c->tmp[r]) (void)hipFree(c->tmp[r]);
            (void)scv_destroy(c->ctx[r]);
        }
        delete c;
        return SCV_OK;
    });
}

int scv_comm_create(scv_comm** out, const int* devices, int n, uint32_t ctx_flags, uint32_t comm_flags) {
    return guarded([&]() -> int {
        if (!out) return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create: out is NULL");
        *out = nullptr;
        int visible = 0;
        if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return scv::comm_fail(SCV_ERR_NO_DEVICE, "no HIP device visible");
        if (n == 0 || !devices) n = visible;                       // all visible devices
        if (n < 1 || n > scv::kMaxRanks) return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create: %d ranks (1..%d supported)", n, scv::kMaxRanks);
        scv_comm* c = new (std::nothrow) scv_comm();
        if (!c) return scv::comm_fail(SCV_ERR_ALLOC, "out of host memory");
        c->n = n;
        c->flags = comm_flags;
        DeviceScope scope;
        for (int r = 0; r < n; ++r) {
            const int d = devices ? devices[r] : r;
            if (d < 0 || d >= visible) { scv_comm_destroy(c); return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create: device %d out of range (%d visible)", d, visible); }
            c->devices.push_back(d);
            scv_ctx* x = nullptr;
            if (int rc = scv_create(&x, d, ctx_flags)) { scv_comm_destroy(c); return rc; }
            c->ctx.push_back(x);
        }
        c->ready.assign(n, nullptr); c->done.assign(n, nullptr); c->tmp.assign(n, nullptr);
        for (int r = 0; r < n; ++r) {
            if (hipSetDevice(c->devices[r]) != hipSuccess || hipEventCreateWithFlags(&c->ready[r], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->done[r], hipEventDisableTiming) != hipSuccess) {
                scv_comm_destroy(c);
                return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create: event creation failed on device %d", c->devices[r]);
            }
            for (int j = 0; j < n; ++j) {                           // every rank reads every other rank's buffer: peer access both ways
                if (c->devices[j] == c->devices[r]) continue;
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, c->devices[r], c->devices[j]);
                if (!can && !(comm_flags & SCV_COMM_RCCL)) {
                    scv_comm_destroy(c);
                    return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create: device %d cannot access device %d (no xGMI / PCIe peer path): use SCV_COMM_RCCL", c->devices[r], c->devices[j]);
                }
                const hipError_t e = hipDeviceEnablePeerAccess(c->devices[j], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled && !(comm_flags & SCV_COMM_RCCL)) {
                    scv_comm_destroy(c);
                    return scv::comm_fail(-(int)e, "hipDeviceEnablePeerAccess(%d -> %d): %s", c->devices[r], c->devices[j], hipGetErrorString(e));
                }
                (void)hipGetLastError();
            }
        }
        if (comm_flags & SCV_COMM_RCCL) {
            if (!g_rccl.load()) {
                const char* why = dlerror();
                scv_comm_destroy(c);
                return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create: librccl could not be loaded (%s)", why ? why : "symbols missing");
            }
            c->nccl.assign(n, nullptr);
            const int rc = g_rccl.CommInitAll(c->nccl.data(), n, c->devices.data());
            if (rc != 0) {
                const char* msg = g_rccl.GetErrorString(rc);
                c->nccl.clear();
                scv_comm_destroy(c);
                return scv::comm_fail(-1000 - rc, "ncclCommInitAll: %s", msg);
            }
        }
        if (n > 1 || (comm_flags & SCV_COMM_RCCL) || peer_fault_for_test()) {
            for (int r = 0; r < n; ++r) {
                if (hipSetDevice(c->devices[r]) != hipSuccess || hipMalloc(&c->tmp[r], kTmpBytesAtCreate) != hipSuccess) {
                    scv_comm_destroy(c);
                    return scv::comm_fail(SCV_ERR_ALLOC, "scv_comm_create: staging buffer on device %d", c->devices[r]);
                }
            }
            c->tmp_bytes = kTmpBytesAtCreate;
            // First contact with the devices happens HERE, loudly: known patterns through every peer path and through one whole
            // all-reduce, verified on every device -- wrong xGMI visibility is an error at create, not a wrong accuracy later.
            if (int rc = comm_selftest(c)) { scv_comm_destroy(c); return rc; }
        }
        *out = c;
        return SCV_OK;
    });
}

int scv_comm_size(const scv_comm* c) { return c ? c->n : 0; }

scv_ctx* scv_comm_ctx(scv_comm* c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->ctx[rank] : nullptr; }

int scv_allreduce_counters(scv_comm* c, int64_t* const* buffers, int64_t count) {
    return guarded([&]() -> int {
        if (!c || !buffers) return scv::comm_fail(SCV_ERR_ARG, "scv_allreduce_counters: NULL argument");
        if (count < 0) return scv::comm_fail(SCV_ERR_ARG, "scv_allreduce_counters: negative count");
        for (int r = 0; r < c->n; ++r)
            if (!buffers[r] && count > 0) return scv::comm_fail(SCV_ERR_ARG, "scv_allreduce_counters: buffer of rank %d is NULL", r);
        if (count == 0 || (c->n == 1 && !(c->flags & SCV_COMM_RCCL))) return SCV_OK;    // one rank: the buffer already holds the sum
        DeviceScope scope;
        if (c->flags & SCV_COMM_RCCL) {                            // (one rank included: a 1-GPU box then executes the RCCL call path itself)
            int rc = g_rccl.GroupStart();
            for (int r = 0; r < c->n && rc == 0; ++r)
                rc = g_rccl.AllReduce(buffers[r], buffers[r], (size_t)count, kNcclInt64, kNcclSum, c->nccl[r], scv::ctx_stream(c->ctx[r]));
            const int rc2 = g_rccl.GroupEnd();
            if (rc == 0) rc = rc2;
            if (rc != 0) return scv::comm_fail(-1000 - rc, "ncclAllReduce: %s", g_rccl.GetErrorString(rc));
            return SCV_OK;
        }
        // ---- one-shot over peer access --------------------------------------------------------------------------------
        const size_t bytes = (size_t)count * sizeof(int64_t);
        if (bytes > c->tmp_bytes) {
            for (int r = 0; r < c->n; ++r) {                       // growing synchronises, frees and allocates: not while a stream is being captured
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(scv::ctx_stream(c->ctx[r]), &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
                if (cap != hipStreamCaptureStatusNone)
                    return scv::comm_fail(SCV_ERR_ARG, "scv_allreduce_counters: %lld words need a larger staging buffer while rank %d's stream is being captured: "
                                          "run this size once outside the capture first", (long long)count, r);
            }
            for (int r = 0; r < c->n; ++r) {
                COMM_HIP(hipSetDevice(c->devices[r]));
                COMM_HIP(hipStreamSynchronize(scv::ctx_stream(c->ctx[r])));
                if (c->tmp[r]) { COMM_HIP(hipFree(c->tmp[r])); c->tmp[r] = nullptr; }
                COMM_HIP(hipMalloc(&c->tmp[r], bytes));
            }
            c->tmp_bytes = bytes;
        }
        scv::PeerPtrs in;
        for (int r = 0; r < scv::kMaxRanks; ++r) in.p[r] = r < c->n ? reinterpret_cast<const long long*>(buffers[r]) : nullptr;
        for (int r = 0; r < c->n; ++r) {                           // 1. every rank's counters are complete at `ready`
            COMM_HIP(hipSetDevice(c->devices[r]));
            COMM_HIP(hipEventRecord(c->ready[r], scv::ctx_stream(c->ctx[r])));
        }
        for (int r = 0; r < c->n; ++r) {                           // 2. sum all ranks' buffers into this rank's staging buffer
            COMM_HIP(hipSetDevice(c->devices[r]));
            hipStream_t s = scv::ctx_stream(c->ctx[r]);
            for (int j = 0; j < c->n; ++j) if (j != r) COMM_HIP(hipStreamWaitEvent(s, c->ready[j], 0));
            int64_t grid = (count + 255) / 256;
            if (grid > 64) grid = 64;
            if (c->peer_loads == 1) hipLaunchKernelGGL(scv::scv_sum_peers_k<false>, dim3((unsigned)grid), dim3(256), 0, s, in, c->n, static_cast<long long*>(c->tmp[r]), count);
            else hipLaunchKernelGGL(scv::scv_sum_peers_k<true>, dim3((unsigned)grid), dim3(256), 0, s, in, c->n, static_cast<long long*>(c->tmp[r]), count);
            COMM_HIP(hipGetLastError());
            COMM_HIP(hipEventRecord(c->done[r], s));
        }
        for (int r = 0; r < c->n; ++r) {                           // 3. nobody reads the buffers any more: the sum replaces them
            COMM_HIP(hipSetDevice(c->devices[r]));
            hipStream_t s = scv::ctx_stream(c->ctx[r]);
            for (int j = 0; j < c->n; ++j) if (j != r) COMM_HIP(hipStreamWaitEvent(s, c->done[j], 0));
            COMM_HIP(hipMemcpyAsync(buffers[r], c->tmp[r], bytes, hipMemcpyDeviceToDevice, s));
        }
        return SCV_OK;
    });
}

// In-place all-gather of byte blocks: bufs[r] (on rank r's device) is the WHOLE gathered buffer, in which rank r has written its
// own block [off_r, off_r + bytes[r]), off_r = bytes[0] + ... + bytes[r - 1]; afterwards every rank's buffer holds every block.
static int allgather_blocks(scv_comm* c, void* const* bufs, const int64_t* bytes, const char* who) {
    if (!c || !bufs || !bytes) return scv::comm_fail(SCV_ERR_ARG, "%s: NULL argument", who);
    int64_t total = 0;
    std::vector<int64_t> off(c->n);
    for (int r = 0; r < c->n; ++r) {
        if (bytes[r] < 0) return scv::comm_fail(SCV_ERR_ARG, "%s: negative block size of rank %d", who, r);
        off[r] = total;
        total += bytes[r];
    }
    for (int r = 0; r < c->n; ++r)
        if (!bufs[r] && total > 0) return scv::comm_fail(SCV_ERR_ARG, "%s: buffer of rank %d is NULL", who, r);
    if (total == 0 || (c->n == 1 && !(c->flags & SCV_COMM_RCCL))) return SCV_OK;
    DeviceScope scope;
    if (c->flags & SCV_COMM_RCCL) {                            // one broadcast per block, all in one group
        int rc = g_rccl.GroupStart();
        for (int root = 0; root < c->n && rc == 0; ++root) {
            if (bytes[root] == 0) continue;
            for (int r = 0; r < c->n && rc == 0; ++r) {
                char* blk = static_cast<char*>(bufs[r]) + off[root];
                rc = g_rccl.Broadcast(blk, blk, (size_t)bytes[root], kNcclInt8, root, c->nccl[r], scv::ctx_stream(c->ctx[r]));
            }
        }
        const int rc2 = g_rccl.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) return scv::comm_fail(-1000 - rc, "%s: ncclBroadcast: %s", who, g_rccl.GetErrorString(rc));
        return SCV_OK;
    }
    for (int r = 0; r < c->n; ++r) {                           // 1. every rank's own block is complete at `ready`
        COMM_HIP(hipSetDevice(c->devices[r]));
        COMM_HIP(hipEventRecord(c->ready[r], scv::ctx_stream(c->ctx[r])));
    }
    for (int r = 0; r < c->n; ++r) {                           // 2. every rank PULLS the other ranks' blocks over its own links
        COMM_HIP(hipSetDevice(c->devices[r]));
        hipStream_t s = scv::ctx_stream(c->ctx[r]);
        for (int j = 0; j < c->n; ++j) {
            if (j == r || bytes[j] == 0) continue;
            COMM_HIP(hipStreamWaitEvent(s, c->ready[j], 0));
            COMM_HIP(hipMemcpyAsync(static_cast<char*>(bufs[r]) + off[j], static_cast<const char*>(bufs[j]) + off[j], (size_t)bytes[j], hipMemcpyDeviceToDevice, s));
        }
        COMM_HIP(hipEventRecord(c->done[r], s));
    }
    for (int r = 0; r < c->n; ++r) {                           // 3. a rank's later work may overwrite its block only when every reader has it
        COMM_HIP(hipSetDevice(c->devices[r]));
        hipStream_t s = scv::ctx_stream(c->ctx[r]);
        for (int j = 0; j < c->n; ++j) if (j != r) COMM_HIP(hipStreamWaitEvent(s, c->done[j], 0));
    }
    return SCV_OK;
}

int scv_allgather_cells(scv_comm* c, scv_cell* const* tables, const int64_t* rows, int32_t B) {
    return guarded([&]() -> int {
        if (!c || !rows || B < 0) return scv::comm_fail(SCV_ERR_ARG, "scv_allgather_cells: bad argument");
        std::vector<int64_t> bytes(c->n);
        for (int r = 0; r < c->n; ++r) {
            if (rows[r] < 0) return scv::comm_fail(SCV_ERR_ARG, "scv_allgather_cells: negative row count of rank %d", r);
            bytes[r] = rows[r] * (int64_t)B * (int64_t)sizeof(scv_cell);
        }
        return allgather_blocks(c, reinterpret_cast<void* const*>(tables), bytes.data(), "scv_allgather_cells");
    });
}

int scv_allgather_i64(scv_comm* c, int64_t* const* buffers, const int64_t* counts) {
    return guarded([&]() -> int {
        if (!c || !counts) return scv::comm_fail(SCV_ERR_ARG, "scv_allgather_i64: bad argument");
        std::vector<int64_t> bytes(c->n);
        for (int r = 0; r < c->n; ++r) {
            if (counts[r] < 0) return scv::comm_fail(SCV_ERR_ARG, "scv_allgather_i64: negative count of rank %d", r);
            bytes[r] = counts[r] * (int64_t)sizeof(int64_t);
        }
        return allgather_blocks(c, reinterpret_cast<void* const*>(buffers), bytes.data(), "scv_allgather_i64");
    });
}

int scv_comm_get_stat(scv_comm* c, const char* key, int64_t* out) {
    return guarded([&]() -> int {
        if (!c || !key || !out) return scv::comm_fail(SCV_ERR_ARG, "scv_comm_get_stat: NULL argument");
        if (!strcmp(key, "selftest_words")) *out = c->stat_selftest_words;
        else if (!strcmp(key, "staging_bytes")) *out = (int64_t)c->tmp_bytes;
        else if (!strcmp(key, "peer_loads")) *out = c->peer_loads;
        else if (!strcmp(key, "selftest_nt_ok")) *out = c->selftest_nt_ok;
        else if (!strcmp(key, "selftest_plain_ok")) *out = c->selftest_plain_ok;
        else return scv::comm_fail(SCV_ERR_ARG, "scv_comm_get_stat: unknown key '%s'", key);
        return SCV_OK;
    });
}

int scv_comm_sync(scv_comm* c) {
    return guarded([&]() -> int {
        if (!c) return scv::comm_fail(SCV_ERR_ARG, "comm is NULL");
        int first = SCV_OK;
        for (int r = 0; r < c->n; ++r) {
            const int rc = scv_sync(c->ctx[r]);
            if (rc != SCV_OK && first == SCV_OK) first = rc;
        }
        return first;
    });
}

}  // extern "C"

namespace {
// Create-time self-test (ranks > 1, or RCCL).  Two rounds with different patterns -- the second round catches a reader that kept
// stale lines of a peer's buffer from the first:
//   (a) SCV_COMM_PEER: every rank reads every OTHER rank's pattern buffer directly, ordered by the same cross-device events as the
//       all-reduce, once with nontemporal loads (the access path of scv_sum_peers_k) and once with ordinary loads, and compares it
//       with that rank's closed-form pattern.  The result decides how this communicator reads its peers ("peer_loads": nontemporal
//       when that was right in every round, else ordinary loads when THOSE were right in every round, else the create fails and
//       names the device pair -- the caller then uses SCV_COMM_RCCL; MultiDeviceEngine does so by itself);
//   (b) one scv_allreduce_counters of the pattern buffers, verified on every device against the closed-form sum -> names the device;
//   (c) one scv_allgather_i64 of per-rank blocks, verified on every device.
int comm_selftest(scv_comm* c) {
    constexpr int64_t K = 8216 + 1;                            // the production payload: packed counters at B = 8 + the error word
    const int n = c->n;
    const bool peer = !(c->flags & SCV_COMM_RCCL);
    const long long skew = (peer && peer_fault_for_test()) ? 1 : 0;
    std::vector<long long*> buf(n, nullptr), gat(n, nullptr);
    std::vector<unsigned long long*> res(n, nullptr);
    auto cleanup = [&]() {
        for (int r = 0; r < n; ++r) {
            (void)hipSetDevice(c->devices[r]);
            if (buf[r]) (void)hipFree(buf[r]);
            if (gat[r]) (void)hipFree(gat[r]);
            if (res[r]) (void)hipFree(res[r]);
        }
    };
    struct Cleanup { decltype(cleanup)& f; ~Cleanup() { f(); } } guard{cleanup};
    DeviceScope scope;
    const int64_t blk = 1031;                                  // all-gather block of every rank (words)
    // result slots of a rank (2 words each: wrong words, first wrong word): [0, n) pair reads, nontemporal | [n, 2n) pair reads,
    // ordinary loads | 2n the all-reduce | 2n + 1 the all-gather
    const int slots = 2 * n + 2;
    const size_t res_bytes = (size_t)(2 * slots) * sizeof(unsigned long long);
    for (int r = 0; r < n; ++r) {
        COMM_HIP(hipSetDevice(c->devices[r]));
        COMM_HIP(hipMalloc((void**)&buf[r], K * sizeof(long long)));
        COMM_HIP(hipMalloc((void**)&gat[r], (size_t)n * blk * sizeof(long long)));
        COMM_HIP(hipMalloc((void**)&res[r], res_bytes));
    }
    std::vector<unsigned long long> init(2 * slots);
    for (size_t i = 0; i < init.size(); i += 2) { init[i] = 0; init[i + 1] = ~0ull; }
    std::vector<unsigned long long> host(2 * slots);
    bool nt_ok = true, plain_ok = true;
    char nt_msg[384] = "";
    for (int round = 0; round < 2; ++round) {
        for (int r = 0; r < n; ++r) {
            COMM_HIP(hipSetDevice(c->devices[r]));
            hipStream_t s = scv::ctx_stream(c->ctx[r]);
            COMM_HIP(hipMemcpyAsync(res[r], init.data(), res_bytes, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(scv::scv_comm_fill_k, dim3(8), dim3(256), 0, s, buf[r], r, round, K);
            hipLaunchKernelGGL(scv::scv_comm_fill_k, dim3(8), dim3(256), 0, s, gat[r] + (int64_t)r * blk, r, round + 2, blk);
            COMM_HIP(hipGetLastError());
            COMM_HIP(hipEventRecord(c->ready[r], s));
        }
        if (peer && n > 1) {                                   // (a) pairwise peer reads, both load kinds
            for (int r = 0; r < n; ++r) {
                COMM_HIP(hipSetDevice(c->devices[r]));
                hipStream_t s = scv::ctx_stream(c->ctx[r]);
                for (int j = 0; j < n; ++j) {
                    if (j == r) continue;
                    COMM_HIP(hipStreamWaitEvent(s, c->ready[j], 0));
                    hipLaunchKernelGGL(scv::scv_comm_verify_k<true>, dim3(8), dim3(256), 0, s, buf[j], j, j + 1, round, K, res[r] + 2 * j, skew);
                    hipLaunchKernelGGL(scv::scv_comm_verify_k<false>, dim3(8), dim3(256), 0, s, buf[j], j, j + 1, round, K, res[r] + 2 * (n + j), skew);
                    COMM_HIP(hipGetLastError());
                }
                COMM_HIP(hipEventRecord(c->done[r], s));
            }
            for (int r = 0; r < n; ++r) {                      // nobody overwrites its buffer (the all-reduce below) before the readers are done
                COMM_HIP(hipSetDevice(c->devices[r]));
                for (int j = 0; j < n; ++j) if (j != r) COMM_HIP(hipStreamWaitEvent(scv::ctx_stream(c->ctx[r]), c->done[j], 0));
            }
            // the verdict on the two load kinds is needed before the all-reduce below picks one (a host sync at create costs nothing)
            for (int r = 0; r < n; ++r) {
                COMM_HIP(hipSetDevice(c->devices[r]));
                COMM_HIP(hipMemcpyAsync(host.data(), res[r], res_bytes, hipMemcpyDeviceToHost, scv::ctx_stream(c->ctx[r])));
                COMM_HIP(hipStreamSynchronize(scv::ctx_stream(c->ctx[r])));
                for (int j = 0; j < n; ++j) {
                    if (host[2 * j] && nt_ok) {
                        nt_ok = false;
                        snprintf(nt_msg, sizeof nt_msg, "round %d: device %d (rank %d) reads %llu of %lld words of device %d's (rank %d) buffer wrong over peer access "
                                 "(nontemporal loads), first at word %llu", round, c->devices[r], r, host[2 * j], (long long)K, c->devices[j], j, host[2 * j + 1]);
                    }
                    if (host[2 * (n + j)]) plain_ok = false;
                }
            }
            c->selftest_nt_ok = nt_ok ? 1 : 0;
            c->selftest_plain_ok = plain_ok ? 1 : 0;
            if (!nt_ok && !plain_ok)
                return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create self-test (%s; ordinary loads wrong too): peer visibility between this device pair is broken -- "
                                      "use SCV_COMM_RCCL", nt_msg);
            c->peer_loads = nt_ok ? 0 : 1;
        }
        std::vector<int64_t*> bp(n), gp(n);
        std::vector<int64_t> cnt(n, blk);
        for (int r = 0; r < n; ++r) { bp[r] = reinterpret_cast<int64_t*>(buf[r]); gp[r] = reinterpret_cast<int64_t*>(gat[r]); }
        if (int rc = scv_allreduce_counters(c, bp.data(), K)) return rc;                 // (b)
        if (int rc = scv_allgather_i64(c, gp.data(), cnt.data())) return rc;             // (c)
        for (int r = 0; r < n; ++r) {
            COMM_HIP(hipSetDevice(c->devices[r]));
            hipStream_t s = scv::ctx_stream(c->ctx[r]);
            hipLaunchKernelGGL(scv::scv_comm_verify_k<true>, dim3(8), dim3(256), 0, s, buf[r], 0, n, round, K, res[r] + 2 * (2 * n), skew);
            for (int j = 0; j < n; ++j)                        // block j of the gathered buffer = rank j's pattern (round + 2); one slot for all blocks
                hipLaunchKernelGGL(scv::scv_comm_verify_k<true>, dim3(8), dim3(256), 0, s, gat[r] + (int64_t)j * blk, j, j + 1, round + 2, blk, res[r] + 2 * (2 * n + 1), skew);
            COMM_HIP(hipGetLastError());
        }
        for (int r = 0; r < n; ++r) {
            COMM_HIP(hipSetDevice(c->devices[r]));
            COMM_HIP(hipMemcpyAsync(host.data(), res[r], res_bytes, hipMemcpyDeviceToHost, scv::ctx_stream(c->ctx[r])));
            COMM_HIP(hipStreamSynchronize(scv::ctx_stream(c->ctx[r])));
            const char* how = peer ? (c->peer_loads == 1 ? "one-shot over peer access, ordinary loads" : "one-shot over peer access") : "RCCL";
            if (host[2 * (2 * n)])
                return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create self-test (round %d): after one all-reduce of known patterns device %d (rank %d) holds %llu of %lld wrong sums "
                                      "(first at word %llu; %s)", round, c->devices[r], r, host[2 * (2 * n)], (long long)K, host[2 * (2 * n) + 1], how);
            if (host[2 * (2 * n + 1)])
                return scv::comm_fail(SCV_ERR_ARG, "scv_comm_create self-test (round %d): after one all-gather of known blocks device %d (rank %d) holds %llu wrong words "
                                      "(first at word %llu of a block; %s)", round, c->devices[r], r, host[2 * (2 * n + 1)], host[2 * (2 * n + 1) + 1], peer ? "peer copies" : "RCCL");
        }
    }
    c->stat_selftest_words = 2 * (K * (int64_t)(peer ? n : 1) + (int64_t)n * blk);
    return SCV_OK;
}
}  // namespace
