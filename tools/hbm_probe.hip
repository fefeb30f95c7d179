// hbm_probe.hip -- measurement utility (not product code): attainable HBM READ bandwidth on gfx950
// for the access-pattern class of scv_hist_argmax (every workgroup streams whole contiguous 4 MiB
// cells with 16-byte loads), so roofline.frac can also be quoted against a measured read ceiling
// instead of the 8 TB/s datasheet number -- and the known-bytes kernel on which the gfx950
// FETCH_SIZE half-count is calibrated (MI355X_MICROARCH.md, HBM section).
//
//   hbm_probe.bin [cells=10000]          sweep of read variants over cells x 4 MiB (10000 = the bench chunk)
//   hbm_probe.bin [cells] --quick        the best four variants only (bench.py runs this beside its timed region)
//   hbm_probe.bin [cells] --short        what wave-per-batch kernels can pull: 2-8 KiB steps, 4-32 waves per CU
//   hbm_probe.bin [cells] --percu        what ONE workgroup (one CU) can pull, and how that adds up: grids of 1, 4, 16, 64, 128, 256
//                                        workgroups of 1024 threads, each streaming its own 256 KiB / 512 KiB / 4 MiB segment ONCE
//                                        (the few-huge-cells shapes: split-N segments, C2's 512 KiB cells) -- GB/s per workgroup
//                                        and in total, and the time one segment takes (the floor of those shapes)
//   hbm_probe.bin [cells] --c2           the floor of BASELINE config 2: a pure read of its 125.8 MB in ONE launch, cut into 240 ... 1920 segments
//   hbm_probe.bin [cells] --dma          the ceiling of the LDS-DMA input path of scv_sort_cells: waves that only copy 2-16 KiB blocks HBM -> LDS
//   hbm_probe.bin [cells] --dmawork      ... with the VALU work of a sorted-cells step behind the copy: every wave copying for itself against one
//                                        producer wave per workgroup (the design question of scv_sort_cells, round 4)
//   hbm_probe.bin [cells] --vmemq        cycles a wave spends issuing its q-th LDS-DMA piece, back to back: where the memory pipe's queue is full
//   hbm_probe.bin [cells] --calib        3 launches of ONE variant (read_cells_pipe<4>, grid 250 x 1024) and nothing
//                                        else: run under `rocprofv3 --pmc FETCH_SIZE` to get FETCH_SIZE per launch
//                                        for exactly cells * 4 MiB of algorithmic reads
//
// Variants:
//   cells      the round-1 probe: per cell, U loads then U consumes; no load is in flight while a wave consumes
//   pipe       register double buffering: the U loads of step i+1 are issued BEFORE step i is consumed, across
//              cell boundaries too (what the vote kernel's cross-cell prefetch + 16 de-synchronised waves achieve)
//   stride     plain grid-stride loop over the whole buffer (no cell structure)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ v4i ld(const v4i* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int U, bool NT>
__global__ void read_cells(const v4i* __restrict__ src, long cell_vecs, long ncells, int* sink) {
    int acc = 0;
    for (long cell = blockIdx.x; cell < ncells; cell += gridDim.x) {
        const v4i* p = src + cell * cell_vecs;
        for (long i = threadIdx.x; i + (long)(U - 1) * blockDim.x < cell_vecs; i += (long)U * blockDim.x) {
            v4i x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld<NT>(p + i + (long)u * blockDim.x);
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
        }
    }
    if (acc == 0x12345678) *sink = acc;
}

// The workgroup's cells form one logical stream of steps (U*T vectors each); step s+1 is loaded before
// step s is consumed, so every lane always has U..2U loads in flight.
template <int U, bool NT>
__global__ void read_cells_pipe(const v4i* __restrict__ src, long cell_vecs, long ncells, int* sink) {
    int acc = 0;
    const long T = blockDim.x;
    const long steps_per_cell = cell_vecs / (U * T);           // host guarantees divisibility
    const long my_cells = blockIdx.x < ncells ? (ncells - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long nsteps = my_cells * steps_per_cell;
    auto addr = [&](long s) {
        const long c = s / steps_per_cell, k = s - c * steps_per_cell;
        return src + (blockIdx.x + c * gridDim.x) * cell_vecs + k * U * T + threadIdx.x;
    };
    v4i cur[U], nxt[U];
    if (nsteps > 0) {
        const v4i* p = addr(0);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = ld<NT>(p + (long)u * T);
    }
    for (long s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) {
            const v4i* p = addr(s + 1);
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = ld<NT>(p + (long)u * T);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    if (acc == 0x12345678) *sink = acc;
}

template <int U, bool NT>
__global__ void read_gridstride(const v4i* __restrict__ src, long nvec, int* sink) {
    int acc = 0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < nvec; i += U * stride) {
        v4i x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld<NT>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    if (acc == 0x12345678) *sink = acc;
}

// LDS-DMA stream (the input path of scv_sort_cells): every wave copies consecutive blocks of `pieces` KiB HBM -> LDS with
// global_load_lds_dwordx4 (1 KiB per instruction), waits for the block, reads one word of it and goes on -- nothing else.
// `pad`: every 17th 16-byte slot re-reads the slot before it (the padded image of 64-vote rows: 17 pieces per 16 KiB block).
__global__ void dma_blocks(const char* __restrict__ src, long nblocks, int pieces, int pad, int nt, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned region = (unsigned)pieces * 1024u + (pad ? 1024u : 0u);
    const unsigned rbase = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned*)lds_dyn + (unsigned)wid * region));
    const long wave = (long)blockIdx.x * nw + __builtin_amdgcn_readfirstlane(wid), nwaves = (long)gridDim.x * nw;
    const int np = pieces + (pad ? 1 : 0);
    int acc = 0;
    for (long b = wave; b < nblocks; b += nwaves) {
        const char* g = src + b * (long)pieces * 1024;
        for (int q = 0; q < np; ++q) {
            unsigned s = (unsigned)q * 64u + (unsigned)lane;                 // slot of the image
            unsigned off = pad ? ((s / 17u) * 16u + (s % 17u < 16u ? s % 17u : 15u)) * 16u : s * 16u;
            if (off > (unsigned)pieces * 1024u - 16u) off = (unsigned)pieces * 1024u - 16u;
            unsigned keep;
            const unsigned dst = rbase + (unsigned)q * 1024u;
            if (nt) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(off), "s"(g), "s"(dst) : "memory");
            else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(off), "s"(g), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= (int)lds_dyn[(rbase - (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned*)lds_dyn) / 4u + (unsigned)lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678) *sink = acc;
}

// ---- LDS-DMA stream WITH the work of a sorted-cells step behind it (--dmawork): who should issue the copy? ------------------------------
// A wave of scv_sort_cells stages a block of PIECES KiB, reads it into 4 * PIECES registers per lane, and runs ~21 VALU instructions per
// vote on them.  Its copy is "asynchronous" only after the memory pipe has ACCEPTED a piece: with every wave of a CU issuing, a piece waits
// in the pipe's queue and the wave with it (profiles/r04_sort_timeline*.log: 40-58 % of a wave's cycles).  Two forms of the same work:
//   mode 0  every wave copies for itself: wait for the block | rows -> registers | issue the next block's pieces | compute
//   mode 1  NC consumer waves + ONE producer wave per workgroup: the producer issues every piece (it alone waits in the queue) and
//           publishes "block landed" through an LDS flag per consumer once vmcnt says so (vmcnt retires in order: with D requests of
//           np pieces in flight, vmcnt((D - 1) np) means the oldest has landed); a consumer publishes "rows read" the same way
typedef unsigned __attribute__((address_space(3))) lds_uint;
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20) W(21)
        W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41)
        W(42) W(43) W(44) W(45) W(46) W(47) W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60) W(61)
        W(62)
#undef W
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}
__device__ __forceinline__ unsigned flag_load(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void flag_store(unsigned addr, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void dma_piece(const char* g, unsigned off, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(off), "s"(g), "s"(dst) : "memory");
}

template <int PIECES>
__global__ void __launch_bounds__(1024) dma_work(const char* __restrict__ src, long nblocks, int k_iter, int mode, int depth, int nb, int npd, int* sink) {
    constexpr int NR = 4 * PIECES;                                   // registers of a lane's row
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int nc = mode ? nw - npd : nw;                             // consumer waves (mode 1: the last npd waves are producers)
    const unsigned region = (unsigned)PIECES * 1024u + 1024u;        // one buffer (+ one pad piece, as the 64-vote image)
    const unsigned lds0 = (unsigned)(unsigned long)(lds_uint*)lds_dyn;
    const unsigned flags = lds0 + (unsigned)(nc * nb) * region;      // landed[nc] | consumed[nc]: blocks of consumer c that have landed / been read
    if (threadIdx.x < 2 * nc) lds_dyn[(flags - lds0) / 4 + threadIdx.x] = 0;
    __syncthreads();
    const long nwaves = (long)gridDim.x * nc;
    constexpr int np = PIECES + 1;
    auto issue = [&](int c, long k) {                                // block k of consumer c into buffer k % nb
        const long b = (long)blockIdx.x * nc + c + k * nwaves;
        const char* g = src + b * (long)PIECES * 1024;
        const unsigned rb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(c * nb + (int)(k % nb)) * region));
#pragma unroll
        for (int q = 0; q < np; ++q) {
            unsigned off = ((unsigned)q * 64u + (unsigned)lane) * 16u;
            if (off > (unsigned)PIECES * 1024u - 16u) off = (unsigned)PIECES * 1024u - 16u;
            dma_piece(g, off, rb + (unsigned)q * 1024u);
        }
    };
    auto blocks_of = [&](int c) -> long {                            // how many blocks consumer c processes
        const long first = (long)blockIdx.x * nc + c;
        return first < nblocks ? (nblocks - first + nwaves - 1) / nwaves : 0;
    };
    if (mode && wid >= nc) {
        // ---- a producer: serves the consumers c = pid, pid + npd, ...; lane c holds consumer c's counters.  One ds_read shows every
        // consumer's progress; whoever has a free buffer and blocks left gets its next block issued; requests land in issue order.
        __builtin_amdgcn_s_setprio(3);
        const int pid = wid - nc;
        const bool mine = lane < nc && lane % npd == pid;
        const unsigned total = mine ? (unsigned)blocks_of(lane) : 0u;
        unsigned issued = 0, landed = 0;                             // (per lane = per consumer)
        unsigned ring = 0;                                           // lane i: consumer of the i-th request in flight (mod 64)
        int head = 0, inflight = 0;
        auto retire = [&]() {
            wait_vmcnt((inflight - 1) * np);
            const int c = __builtin_amdgcn_readlane((int)ring, head);
            head = (head + 1) & 63; --inflight;
            if (lane == c) { ++landed; flag_store(flags + 4u * (unsigned)c, landed); }
        };
        for (;;) {
            unsigned consumed = 0;
            if (mine) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(consumed) : "v"(flags + 4u * (unsigned)(nc + lane)) : "memory");
            const bool more = mine && issued < total;
            unsigned long long ready = __ballot(more && issued < consumed + (unsigned)nb);
            if (!__ballot(more) && !inflight) break;
            if (!ready) { if (inflight) retire(); else __builtin_amdgcn_s_sleep(1); continue; }
            while (ready) {
                const int c = __builtin_ctzll(ready);
                ready &= ready - 1;
                if (inflight == depth) retire();
                const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)issued, c);
                issue(c, (long)k);
                if (lane == c) ++issued;
                if (lane == ((head + inflight) & 63)) ring = (unsigned)c;
                ++inflight;
            }
        }
        return;
    }
    // ---- a consumer (mode 0: it also copies for itself, one buffer)
    const int c = wid;
    unsigned acc = 0;
    const long nk = blocks_of(c);
    if (!mode && nk > 0) issue(c, 0);
    for (long k = 0; k < nk; ++k) {
        if (mode) { while (flag_load(flags + 4u * (unsigned)c) < (unsigned)k + 1u) __builtin_amdgcn_s_sleep(1); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned rb = lds0 + (unsigned)(c * nb + (int)(k % nb)) * region;
        unsigned r[NR];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            // lane l's row: PIECES slots at slot stride PIECES + 1 (odd: conflict-free)
            asm volatile("ds_read_b128 %0, %1" : "=v"(*(v4i*)&r[4 * i]) : "v"(rb + ((unsigned)lane * (unsigned)(PIECES | 1) + (unsigned)i) * 16u) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mode) flag_store(flags + 4u * (unsigned)(nc + c), (unsigned)k + 1u);
        else if (k + 1 < nk) issue(c, k + 1);
        for (int it = 0; it < k_iter; ++it) {
#pragma unroll
            for (int i = 0; i < NR / 2; ++i) {
                unsigned t;
                asm volatile("v_pk_min_u16 %0, %1, %2\n\tv_pk_max_u16 %2, %1, %2" : "=&v"(t), "+v"(r[i]), "+v"(r[NR - 1 - i]));
                r[i] = t;
            }
#pragma unroll
            for (int i = 0; i + 1 < NR; i += 2) {
                unsigned t;
                asm volatile("v_pk_min_u16 %0, %1, %2\n\tv_pk_max_u16 %2, %1, %2" : "=&v"(t), "+v"(r[i]), "+v"(r[i + 1]));
                r[i] = t;
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) acc ^= r[i];
    }
    if (acc == 0x12345678u) *sink = (int)acc;
}

// ---- how many LDS-DMA pieces does the memory pipe ACCEPT before a wave's next one blocks? (--vmemq) ----------------------------------------
// Every wave issues P pieces back to back (1 KiB each, consecutive source addresses, all into the same 1 KiB of LDS: only the issue is
// measured) and stamps s_memtime around each: cycles[wave][q] = time the q-th issue took.  With one wave on the chip the first pieces are
// accepted in a few dozen cycles each; where the time per issue jumps to a memory latency, the queue in front of the wave is full.
template <int KIND>   // 0: global_load_lds_dwordx4 (LDS-DMA) | 1: global_load_dwordx4 into VGPRs | 2: ... and a ds_write_b128 of an earlier quad after each | 3: the stamp alone | 4: buffer_load_dwordx4 ... lds
__global__ void vmem_queue(const char* __restrict__ src, int pieces, long wave_stride, unsigned* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const long wave = (long)blockIdx.x * nw + wid;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_uint*)lds_dyn + (unsigned)wid * 1024u;
    const char* g = src + wave * wave_stride;
    unsigned long long t = __builtin_readcyclecounter();
    v4i r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = v4i{lane, i, 0, 0};
    for (int q = 0; q < pieces; ++q) {
        if (KIND == 0) dma_piece(g + (long)q * 1024, (unsigned)lane * 16u, (unsigned)__builtin_amdgcn_readfirstlane((int)lds0));
        else if (KIND == 3) { }                                          // the stamp alone
        else if (KIND == 4) {
            // the MUBUF form: descriptor {base, stride 0, 2^32 - 1 bytes, raw 32-bit format} in four SGPRs, byte offset in a VGPR
            const unsigned long long base = (unsigned long long)(g + (long)q * 1024);
            const v4i rs = {(int)(unsigned)base, (int)(unsigned)(base >> 32) & 0xffff, -1, 0x00020000};
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((unsigned)lane * 16u), "s"(rs), "s"((unsigned)__builtin_amdgcn_readfirstlane((int)lds0)) : "memory");
        } else {
            const char* gq = g + (long)q * 1024;
            // (eight destination quads in rotation; the asm is opaque to hipcc: no waits of its own)
            switch (q & 7) {
#define LD(i) case i: asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r[i]) : "v"((unsigned)lane * 16u), "s"(gq) : "memory"); \
                      if (KIND == 2) asm volatile("ds_write_b128 %0, %1" : : "v"(lds0 + (unsigned)lane * 16u), "v"(r[(i + 4) & 7]) : "memory"); break;
                LD(0) LD(1) LD(2) LD(3) LD(4) LD(5) LD(6) LD(7)
#undef LD
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) cycles[wave * (pieces + 1) + q] = (unsigned)(t1 - t);
        t = t1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[wave * (pieces + 1) + pieces] = (unsigned)(t1 - t);
    if (KIND != 0) {
        int acc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
        if (acc == 0x12345678) cycles[0] = (unsigned)acc;
    }
}

__global__ void fill(v4i* dst, long nvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        // pseudo-random words: a constant pattern would flatter the memory system's power budget
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        v4i v = {(int)z, (int)(z >> 32), (int)(z * 3), (int)(z >> 17)};
        dst[i] = v;
    }
}

template <typename F>
double time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    long ncells = 10000;
    bool calib = false, shortcells = false, quick = false, percu = false, dma = false, c2 = false, dmawork = false, vmemq = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--calib")) calib = true;
        else if (!strcmp(argv[i], "--short")) shortcells = true;
        else if (!strcmp(argv[i], "--quick")) quick = true;
        else if (!strcmp(argv[i], "--percu")) percu = true;
        else if (!strcmp(argv[i], "--dma")) dma = true;
        else if (!strcmp(argv[i], "--dmawork")) dmawork = true;
        else if (!strcmp(argv[i], "--vmemq")) vmemq = true;
        else if (!strcmp(argv[i], "--c2")) c2 = true;
        else ncells = atol(argv[i]);
    }
    const long cell_bytes = 4l << 20;
    const long bytes = ncells * cell_bytes, nvec = bytes / 16, cell_vecs = cell_bytes / 16;
    v4i* buf; int* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    fill<<<4096, 256>>>(buf, nvec); CK(hipDeviceSynchronize());
    printf("buffer %.3f GB = %ld cells of 4 MiB (algorithmic read bytes per launch: %ld)\n", bytes / 1e9, ncells, bytes);
    if (calib) {
        for (int r = 0; r < 3; ++r) {
            read_cells_pipe<4, true><<<250, 1024>>>(buf, cell_vecs, ncells, sink);
            CK(hipDeviceSynchronize());
        }
        printf("calib: 3 launches of read_cells_pipe<4,nt> grid 250 x 1024, %ld bytes each\n", bytes);
        return 0;
    }
    if (vmemq) {
        printf("vmemq: cycles a wave spends ISSUING its q-th LDS-DMA piece (global_load_lds_dwordx4, 1 KiB), back to back; median over the waves; last column: the wait for all of them\n");
        constexpr int P = 40;
        unsigned* d_cyc;
        CK(hipMalloc(&d_cyc, sizeof(unsigned) * 256 * 16 * (P + 1)));
        for (int kind : {3, 0, 4, 1, 2})
        for (int blocks : {1, 256}) {
            for (int waves : {1, 4, 8, 16}) {
                const int nwaves = blocks * waves;
                for (int rep = 0; rep < 2; ++rep) {
                    const char* src0 = (const char*)buf + (long)(rep + 2 * kind) * (256l << 20);
                    if (kind == 0) vmem_queue<0><<<blocks, waves * 64, waves * 1024>>>(src0, P, (long)P * 1024, d_cyc);
                    else if (kind == 1) vmem_queue<1><<<blocks, waves * 64, waves * 1024>>>(src0, P, (long)P * 1024, d_cyc);
                    else if (kind == 2) vmem_queue<2><<<blocks, waves * 64, waves * 1024>>>(src0, P, (long)P * 1024, d_cyc);
                    else if (kind == 3) vmem_queue<3><<<blocks, waves * 64, waves * 1024>>>(src0, P, (long)P * 1024, d_cyc);
                    else vmem_queue<4><<<blocks, waves * 64, waves * 1024>>>(src0, P, (long)P * 1024, d_cyc);
                    CK(hipDeviceSynchronize());
                }
                std::vector<unsigned> c((size_t)nwaves * (P + 1));
                CK(hipMemcpy(c.data(), d_cyc, sizeof(unsigned) * c.size(), hipMemcpyDeviceToHost));
                printf("vmemq %-22s %3d workgroup(s) x %2d waves:", kind == 0 ? "global_load_lds_dwordx4" : (kind == 1 ? "global_load_dwordx4" : (kind == 2 ? "load + ds_write_b128" : (kind == 3 ? "(the stamps alone)" : "buffer_load_dwordx4 lds"))), blocks, waves);
                for (int q = 0; q <= P; ++q) {
                    std::vector<unsigned> col(nwaves);
                    for (int w = 0; w < nwaves; ++w) col[w] = c[(size_t)w * (P + 1) + q];
                    std::sort(col.begin(), col.end());
                    printf(q == P ? " | %u" : " %u", col[nwaves / 2]);
                }
                printf("\n");
            }
        }
        return 0;
    }
    if (dmawork) {
        printf("dmawork: the copy + the VALU work of a sorted-cells step (k packed min / max per block and lane, plus up to k / 2 v_mov the compiler keeps: x 1.5 for the "
               "instruction count); self-issued = every wave copies for itself, else producer wave(s) copy for the consumer waves\n");
        auto run = [&](auto kern, int pieces, int nc, int k_valu, int mode, int depth, int nb, int npd) {
            const int nr = 4 * pieces, k_iter = k_valu / (2 * nr);   // one iteration = 2 nr instructions
            const size_t lds = (size_t)nc * nb * ((size_t)pieces * 1024 + 1024) + 8 * (size_t)nc;
            if (lds > 160 * 1024) return;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const long nblocks = bytes / ((long)pieces * 1024);
            const int threads = (nc + (mode ? npd : 0)) * 64;
            if (threads > 1024) return;
            const double ms = time_ms([&] { kern<<<256, threads, lds>>>((const char*)buf, nblocks, k_iter, mode, depth, nb, npd, sink); CK(hipGetLastError()); }, 3);
            if (mode) printf("dmawork %2d KiB blocks x %2d consumer waves x %d buffers  %4d pk_min/pk_max per block  %d producer wave(s), %2d requests in flight each : %7.3f ms = %6.0f GB/s\n",
                             pieces, nc, nb, k_iter * 2 * nr, npd, depth, ms, bytes / ms / 1e6);
            else printf("dmawork %2d KiB blocks x %2d consumer waves x %d buffers  %4d pk_min/pk_max per block  self-issued                                   : %7.3f ms = %6.0f GB/s\n",
                        pieces, nc, nb, k_iter * 2 * nr, ms, bytes / ms / 1e6);
        };
        for (int k_valu : {0, 1024, 1344}) {
            run(dma_work<16>, 16, 8, k_valu, 0, 0, 1, 0);
            run(dma_work<16>, 16, 8, k_valu, 1, 3, 1, 1);
            run(dma_work<16>, 16, 8, k_valu, 1, 3, 1, 2);
            run(dma_work<16>, 16, 4, k_valu, 1, 3, 2, 1);
            run(dma_work<16>, 16, 4, k_valu, 1, 3, 2, 2);
            run(dma_work<16>, 16, 4, k_valu, 1, 2, 2, 2);
        }
        for (int k_valu : {0, 512, 640}) {
            run(dma_work<8>, 8, 16, k_valu, 0, 0, 1, 0);
            run(dma_work<8>, 8, 14, k_valu, 1, 6, 1, 2);
            run(dma_work<8>, 8, 8, k_valu, 1, 6, 2, 1);
            run(dma_work<8>, 8, 8, k_valu, 1, 6, 2, 2);
            run(dma_work<8>, 8, 8, k_valu, 1, 3, 2, 2);
            run(dma_work<8>, 8, 12, k_valu, 1, 3, 1, 2);
        }
        for (int k_valu : {0, 256, 320}) {
            run(dma_work<4>, 4, 16, k_valu, 0, 0, 1, 0);
            run(dma_work<4>, 4, 14, k_valu, 1, 6, 2, 2);
            run(dma_work<4>, 4, 8, k_valu, 1, 6, 4, 2);
            run(dma_work<4>, 4, 8, k_valu, 1, 12, 4, 1);
        }
        for (int k_valu : {0, 128, 160}) {
            run(dma_work<2>, 2, 16, k_valu, 0, 0, 1, 0);
            run(dma_work<2>, 2, 14, k_valu, 1, 10, 3, 2);
            run(dma_work<2>, 2, 8, k_valu, 1, 10, 6, 2);
        }
        return 0;
    }
    if (dma) {
        // the ceiling of the LDS-DMA input path: blocks of 2 / 4 / 8 / 16 KiB per wave (the 8 / 16 / 32 / 64-vote shapes), 8 or 16 waves per CU
        printf("dma: every wave copies consecutive blocks HBM -> LDS (global_load_lds_dwordx4, one buffer per wave, wait, next block); one workgroup per CU\n");
        for (int pieces : {2, 4, 8, 16}) {
            for (int waves : {8, 16}) {
                for (int pad : {0, 1}) {
                    if (pad && pieces != 16) continue;
                    for (int nt : {1, 0}) {
                        const size_t lds = (size_t)waves * ((size_t)pieces * 1024 + (pad ? 1024 : 0));
                        if (lds > 160 * 1024) continue;
                        CK(hipFuncSetAttribute((const void*)dma_blocks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                        const long nblocks = bytes / ((long)pieces * 1024);
                        const double ms = time_ms([&] { dma_blocks<<<256, waves * 64, lds>>>((const char*)buf, nblocks, pieces, pad, nt, sink); CK(hipGetLastError()); });
                        printf("dma     %2d KiB blocks%s x %2d waves per CU %s : %7.3f ms = %6.0f GB/s\n", pieces, pad ? " (padded image)" : "", waves, nt ? "nt   " : "plain", ms, bytes / ms / 1e6);
                    }
                }
            }
        }
        return 0;
    }
    if (c2) {
        // BASELINE config 2 is 30 x 8 cells of 512 KiB = 125.8 MB in ONE launch: what is the floor of a pure read of those bytes, and does it
        // depend on how the launch is cut?  240 / 480 / 960 / 1920 segments x workgroup sizes; U loads of 16 bytes in flight per lane;
        // distinct regions per repetition (nothing served by the 256 MiB Infinity Cache); launch time by hipEvents, median of 40+.
        printf("c2: pure read of 125.8 MB in one launch, by segments x threads per workgroup (us per launch | GB/s)\n");
        const long total_vecs = 240l * 512 * 1024 / 16;
        struct G { int segs, threads, u; };
        for (G g : {G{240, 1024, 4}, G{240, 1024, 8}, G{240, 512, 4}, G{240, 512, 8}, G{480, 1024, 4}, G{480, 512, 4}, G{480, 512, 8}, G{480, 256, 4}, G{960, 512, 4}, G{960, 256, 4},
                    G{960, 256, 8}, G{1920, 256, 4}, G{1920, 128, 4}, G{256, 1024, 4}, G{512, 512, 4}, G{1024, 256, 4}}) {
            const long seg_vecs = total_vecs / g.segs / ((long)g.u * g.threads) * ((long)g.u * g.threads);     // whole steps only (<= 0.6 % fewer bytes)
            const long span = (long)g.segs * seg_vecs;
            const long regions = nvec / span < 64 ? nvec / span : 64;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            std::vector<float> ts;
            for (long r = 0; r < regions; ++r) {
                CK(hipEventRecord(e0));
                if (g.u == 4) read_cells<4, true><<<g.segs, g.threads>>>(buf + r * span, seg_vecs, g.segs, sink);
                else read_cells<8, true><<<g.segs, g.threads>>>(buf + r * span, seg_vecs, g.segs, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            const double ms = ts.empty() ? 0 : ts[ts.size() / 2], best = ts.empty() ? 0 : ts[0];
            printf("c2      %4d segments of %6.1f KiB x %4d threads, U = %d : median %6.1f us (best %6.1f) | %6.0f GB/s\n", g.segs, seg_vecs * 16 / 1024.0, g.threads, g.u,
                   ms * 1e3, best * 1e3, span * 16.0 / ms / 1e6);
        }
        return 0;
    }
    if (percu) {
        // Every workgroup reads ONE segment (grid = number of segments): a launch lasts as long as one CU needs for its segment.
        // Distinct regions of the buffer per repetition (stride below), so nothing is served by the 256 MiB Infinity Cache.
        printf("percu: one 1024-thread workgroup per segment, U = 4 non-temporal 16-byte loads in flight per lane; launch time incl. ~2 us of launch latency\n");
        for (long seg_kib : {256l, 512l, 4096l}) {
            const long seg_vecs = seg_kib * 1024 / 16;
            for (int grid : {1, 4, 16, 64, 128, 256}) {
                const long span = (long)grid * seg_vecs;                       // vectors touched per launch
                const long regions = nvec / span < 64 ? nvec / span : 64;
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                std::vector<float> ts;
                for (long r = 0; r < regions; ++r) {
                    CK(hipEventRecord(e0));
                    read_cells<4, true><<<grid, 1024>>>(buf + r * span, seg_vecs, grid, sink);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                const double ms = ts.empty() ? 0 : ts[ts.size() / 2];
                const double total = span * 16.0 / ms / 1e6;
                printf("percu   segment %5ld KiB x %3d workgroups : %7.1f us per launch | %7.1f GB/s per workgroup | %7.0f GB/s total\n",
                       seg_kib, grid, ms * 1e3, total / grid, total);
            }
        }
        return 0;
    }
    struct Cfg { int grid, threads; };
    if (quick) {
        // the four best variants of the full sweep + the vote kernel's own geometry: what bench.py runs next to its
        // timed region so that kernel and ceiling come from the SAME box (boxes of the pool differ by +-3 %)
        double best = 0;
        auto upd = [&](const char* name, double ms) { const double g = bytes / ms / 1e6; printf("quick   %-28s %6.0f GB/s\n", name, g); if (g > best) best = g; };
        upd("cells U2 nt 512 x 512", time_ms([&] { read_cells<2, true><<<512, 512>>>(buf, cell_vecs, ncells, sink); }, 3));
        upd("cells U2 nt 1000 x 256", time_ms([&] { read_cells<2, true><<<1000, 256>>>(buf, cell_vecs, ncells, sink); }, 3));
        upd("pipe U4 nt 1000 x 256", time_ms([&] { read_cells_pipe<4, true><<<1000, 256>>>(buf, cell_vecs, ncells, sink); }, 3));
        upd("cells U4 nt 250 x 1024", time_ms([&] { read_cells<4, true><<<250, 1024>>>(buf, cell_vecs, ncells, sink); }, 3));
        printf("READ_CEILING_GBPS %.0f\n", best);
        return 0;
    }
    if (shortcells) {
        // what a wave-per-batch kernel can pull: every wave streams 4 KiB (U = 4) or 8 KiB (U = 8) steps of its own,
        // one step in flight behind the one being consumed; G waves per CU as 64- or 256-thread workgroups
        const long cv = 256;                              // 4 KiB "cells"
        const long nc = bytes / (cv * 16);
        for (int wpc : {4, 8, 9, 12, 16, 24, 32}) {
            for (int threads : {64, 256}) {
                const int grid = 256 * wpc * 64 / threads;
                double a4 = time_ms([&] { read_cells_pipe<4, true><<<grid, threads>>>(buf, cv * (threads / 64), nc / (threads / 64), sink); });
                double a8 = time_ms([&] { read_cells_pipe<8, true><<<grid, threads>>>(buf, 2 * cv * (threads / 64), nc / (2 * threads / 64), sink); });
                double a2 = time_ms([&] { read_cells_pipe<2, true><<<grid, threads>>>(buf, cv / 2 * (threads / 64), 2 * nc / (threads / 64), sink); });
                printf("short   %2d waves/CU as %3d-thread WGs : 2 KiB/step %6.0f | 4 KiB/step %6.0f | 8 KiB/step %6.0f GB/s\n", wpc, threads,
                       bytes / a2 / 1e6, bytes / a4 / 1e6, bytes / a8 / 1e6);
            }
        }
        return 0;
    }
    const Cfg cfgs[] = {{250, 1024}, {256, 1024}, {500, 512}, {512, 512}, {1000, 256}, {2000, 256}};
    double best = 0;
    for (const Cfg& c : cfgs) {
        double a = time_ms([&] { read_cells<4, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double b = time_ms([&] { read_cells<4, false><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double d = time_ms([&] { read_cells<8, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double e = time_ms([&] { read_cells<2, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        printf("cells   grid %4d x %4d : U4 nt %6.0f | U4 plain %6.0f | U8 nt %6.0f | U2 nt %6.0f GB/s\n", c.grid, c.threads,
               bytes / a / 1e6, bytes / b / 1e6, bytes / d / 1e6, bytes / e / 1e6);
        best = std::max({best, bytes / a / 1e6, bytes / d / 1e6, bytes / e / 1e6});
    }
    for (const Cfg& c : cfgs) {
        double a = time_ms([&] { read_cells_pipe<4, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double b = time_ms([&] { read_cells_pipe<4, false><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double d = time_ms([&] { read_cells_pipe<2, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        double e = time_ms([&] { read_cells_pipe<8, true><<<c.grid, c.threads>>>(buf, cell_vecs, ncells, sink); });
        printf("pipe    grid %4d x %4d : U4 nt %6.0f | U4 plain %6.0f | U2 nt %6.0f | U8 nt %6.0f GB/s\n", c.grid, c.threads,
               bytes / a / 1e6, bytes / b / 1e6, bytes / d / 1e6, bytes / e / 1e6);
        best = std::max({best, bytes / a / 1e6, bytes / d / 1e6, bytes / e / 1e6});
    }
    const Cfg gs[] = {{256, 1024}, {512, 512}, {2048, 256}, {8192, 256}};
    for (const Cfg& c : gs) {
        double a = time_ms([&] { read_gridstride<4, true><<<c.grid, c.threads>>>(buf, nvec, sink); });
        double b = time_ms([&] { read_gridstride<4, false><<<c.grid, c.threads>>>(buf, nvec, sink); });
        printf("stride  grid %4d x %4d : U4 nt %6.0f | U4 plain %6.0f GB/s\n", c.grid, c.threads, bytes / a / 1e6, bytes / b / 1e6);
        best = std::max(best, bytes / a / 1e6);
    }
    printf("READ_CEILING_GBPS %.0f\n", best);
    return 0;
}
