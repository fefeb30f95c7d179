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

// out[i] = sum over the ranks of in_r[i]; the other ranks' buffers are read through peer access (xGMI)
__global__ __launch_bounds__(256) void scv_sum_peers_k(PeerPtrs in, int n, long long* out, int64_t count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        long long s = 0;
        for (int r = 0; r < n; ++r) s += __builtin_nontemporal_load(in.p[r] + i);
        out[i] = s;
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
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;
constexpr int kNcclInt64 = 4, kNcclSum = 0;     // rccl.h: ncclDataType_t / ncclRedOp_t

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
};

#define COMM_HIP(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return scv::comm_fail(e_ == hipErrorOutOfMemory ? SCV_ERR_ALLOC : -(int)e_, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int scv_comm_destroy(scv_comm* c) {
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
}

int scv_comm_create(scv_comm** out, const int* devices, int n, uint32_t ctx_flags, uint32_t comm_flags) {
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
    *out = c;
    return SCV_OK;
}

int scv_comm_size(const scv_comm* c) { return c ? c->n : 0; }

scv_ctx* scv_comm_ctx(scv_comm* c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->ctx[rank] : nullptr; }

int scv_allreduce_counters(scv_comm* c, int64_t* const* buffers, int64_t count) {
    if (!c || !buffers) return scv::comm_fail(SCV_ERR_ARG, "scv_allreduce_counters: NULL argument");
    if (count < 0) return scv::comm_fail(SCV_ERR_ARG, "scv_allreduce_counters: negative count");
    for (int r = 0; r < c->n; ++r)
        if (!buffers[r] && count > 0) return scv::comm_fail(SCV_ERR_ARG, "scv_allreduce_counters: buffer of rank %d is NULL", r);
    if (count == 0 || c->n == 1) return SCV_OK;                // one rank: the buffer already holds the sum
    DeviceScope scope;
    if (c->flags & SCV_COMM_RCCL) {
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
        hipLaunchKernelGGL(scv::scv_sum_peers_k, dim3((unsigned)grid), dim3(256), 0, s, in, c->n, static_cast<long long*>(c->tmp[r]), count);
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
}

int scv_comm_sync(scv_comm* c) {
    if (!c) return scv::comm_fail(SCV_ERR_ARG, "comm is NULL");
    int first = SCV_OK;
    for (int r = 0; r < c->n; ++r) {
        const int rc = scv_sync(c->ctx[r]);
        if (rc != SCV_OK && first == SCV_OK) first = rc;
    }
    return first;
}

}  // extern "C"
