// scvote_kernels.hip.h -- gfx950 (CDNA4 / MI355X) device code of the self-consistency engine.
//
// scv_hist_argmax     THE hot path: one persistent workgroup streams whole cells (or split-N segments).
//                     Replaces, per (problem, budget) cell, /root/reference/o1.py:181-195 (collect N
//                     votes, sum tokens), o1.py:202 (statistics.multimode) and o1.py:204-213 (tie-aware
//                     score), and accumulates the integer part of the per-budget reduction o1.py:229-245.
// (split cells -- several workgroups per cell -- are merged inside the scv_hist_argmax launch: device atomics + a ticket per cell, round 6)
// scv_few_votes       cells of exactly 1 / 2 / 4 votes in whole blocks (the reference's most common sizes, o1.py:276, 302).
// scv_lane_cells      one lane per cell, registers only: N = 3, 5 .. 7, partial blocks, pool rows of up to 32 votes.
// scv_reg_cells, scv_reg_dense      a cell in the registers of 16 / 32 / 64 lanes, 16-bit bins in LDS (64 < N <= 8192).
// (scv_sort_cells -- one lane per cell, rows by LDS-DMA, packed sorting network: csrc/scvote_sort.hip.h -- 5 / 8 <= N <= 64.)
// scv_lane_prefix, scv_prefix_hist  budgets that are prefixes of one sample pool, one pass: pools of up to 64 votes / beyond 4096
// (scv_prefix_pool -- G lanes per problem, ranks from returning LDS atomics: csrc/scvote_prefix.hip.h -- 65 .. 4096 votes).
// (scv_sort_prefix -- one lane per problem, power-of-two budgets out of one sort: csrc/scvote_sort_prefix.hip.h -- pools of 17 .. 128 votes; 68 .. 128 in two halves: scv_sort_prefix2.)
// scv_reduce_cells    per-budget integer counters from the cell table when cells are short and many.
// scv_bootstrap_k     problem-level bootstrap over the per-cell table.
// scv_synth_fill_k    closed-form synthetic generator (spec: include/scvote.h).
//
// Design (DESIGN.md has the numbers):
//  * Pure integer/indexing work, HBM-bound: 4 algorithmic bytes per vote, nothing written per vote.
//    No MFMA.  The only on-chip resource that can undercut HBM is the LDS atomic unit.
//  * One persistent workgroup streams whole cells: lane l of every wave issues 16-byte
//    (global_load_dwordx4) loads at consecutive addresses, so a wave instruction covers 1 KiB
//    contiguous and U of them are in flight per lane.
//  * The 1024-bin histogram lives in LDS, replicated R times and indexed [bin][lane % R]:
//    word address = bin*R + (lane & (R-1)).  A ds_add_u32 wave instruction is serviced in two
//    32-lane groups over 32 banks; with this layout the bank is (bin*R + lane) % 32, so for R = 32
//    every lane of a group owns its bank (conflict-free for ANY data, including all-equal votes);
//    for R = 16 at most 2 lanes share a bank (free: the 4-cycle issue already covers 2 array
//    cycles), R = 8 -> <= 4-way, R = 4 -> <= 8-way (still above the HBM rate; used for short cells,
//    where LDS footprint -> workgroups per CU -> cells in flight is what matters).  Throughput is therefore independent of the answer distribution:
//    peaked / degenerate inputs (the realistic case: 40-70 % of votes on one bin) cost the same as
//    uniform ones.  This is what a 64-wide wavefront + 160 KiB LDS buys; it is not a warp-shaped
//    design.
//  * Cell epilogue: fold the R copies (ds_read_b128, rotated so the 16-lane groups of a b128 read
//    hit 16 distinct 16-byte slots), zero them in the same pass, then wave-reduce (64 lanes)
//    max / #modes / min-mode, one LDS hop across waves, one 16-byte cell record.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/scvote.h"

namespace scv {

constexpr int kBins = SCV_NUM_BINS;
constexpr int kRedWords = 96;  // cross-wave scratch behind the histogram
constexpr int kSplitTickets = 512;  // split-N in one launch: one arrival counter per split cell (tickets[8 + cell]; a launch splits at most this many cells)

struct AggArgs {
    const int32_t* answers;
    const int32_t* tokens;
    const int32_t* n_valid;
    const int32_t* truth;
    int64_t ncells;
    int64_t N;
    int32_t B;
    scv_cell* cells;
    int64_t* cell_tokens;
    unsigned long long* tie_hits;
    unsigned long long* token_sum;
    unsigned long long* truth_sum;
    uint32_t* err_flag;
    int32_t prefetch;       // != 0: load the first tile of the next item before the current item's epilogue
    int64_t P;              // problems (for the budget-major traversal)
    int32_t sorted;         // != 0: traverse budgets in descending n_valid order
    int32_t segs;           // split-N: segments per cell (1 = whole cells)
    int64_t seg_len;        // split-N: votes per segment
    int32_t wave_lds_words; // register-resident kernels: LDS words per wave (histograms + n_valid cache)
    int32_t pool_rows;      // cell kernels (register-resident, one-lane-per-cell): != 0 = prefix budgets over one pool: cell (p, b) reads
                            // row p of answers / tokens [P, N] (its first n_valid[b] votes) instead of row p * B + b of [P, B, N]
    int32_t acc_classes;    // register-resident kernels: > 0 = per-budget counters accumulate in LDS (this many tie classes per
                            // budget; larger classes go to memory directly) and are flushed once per workgroup
    // single-launch modes of the streaming kernel (agent-scope hand-offs inside the launch, no second kernel):
    uint32_t* tickets;      // [0] workgroups finished | [1] barrier arrivals | [2] barrier generation; [0] and [1] are zero between
                            // launches, [2] only ever grows
    int32_t overwrite;      // != 0: per-budget counters are OVERWRITTEN by the last workgroup to finish (from the cell table)
    unsigned long long* ow_tie;     // overwrite outputs (tie_hits / token_sum / truth_sum are NULL in this mode)
    unsigned long long* ow_tok;
    unsigned long long* ow_truth;
    int32_t boot;           // != 0: after the vote, ALL workgroups meet at a grid barrier and run the bootstrap (one launch)
    int32_t boot_r0, boot_r1, boot_M;
    uint32_t boot_spins;    // grid barrier: polls before a waiting workgroup gives up (error bit 4; the host then runs the bootstrap
                            // as a separate launch at the next scv_sync)
    uint64_t boot_seed;
    unsigned long long* boot_out;   // [r1 - r0][B][M]
    uint32_t* partial;      // split-N: [ncells][1024] the cells' histograms in memory, summed by device atomics (all zero between launches)
    long long* partial_tok; // split-N: [ncells] the cells' token sums, likewise
    int32_t skip_sortable = 0;      // prefix kernels queued BEHIND scv_sort_prefix<NV> (DEVICE mode: the host cannot read n_valid): = NV; the
                                    // launch leaves at once when every budget is of the form that kernel serves (it has done the work)
    int32_t packed_cells = 0;       // SCV_FLAG_PACKED_CELLS: cells is uint32 [P, B] (cells of up to 127 votes; pack_cell below), not scv_cell [P, B]
    int32_t budgets_promised = 0;   // scv_sort_prefix: != 0 = the budgets are KNOWN to be of its form (read by a HOST-mode call, or promised by the
                                    // caller: option prefix_path = 5): a list that is not sets error bit 8 instead of leaving the launch to another kernel
};

// SCV_FLAG_PACKED_CELLS (include/scvote.h): one record in 4 bytes -- max_count | truth_count << 7 | n_modes << 14 | min_mode << 21 | hit << 31; every count
// of a cell of up to 127 votes fits 7 bits; an empty cell (max_count == 0) has min_mode field 1023 and decodes to -1.  The reference's most common
// call is N = 1 (o1.py:302): a 16-byte record per 4-byte vote made the cell table 80 % of that launch's traffic.
__device__ __forceinline__ uint32_t pack_cell(uint32_t maxc, uint32_t tc, uint32_t n_modes, uint32_t mm, uint32_t hit) {
    return (maxc & 0x7fu) | ((tc & 0x7fu) << 7) | ((n_modes & 0x7fu) << 14) | ((maxc ? (mm & 0x3ffu) : 0x3ffu) << 21) | (hit << 31);
}

// 64-lane reductions on the VALU (DPP), not through the LDS crossbar: __shfl_xor lowers to
// ds_bpermute_b32, which has LDS latency and queues behind the histogram atomics.  gfx9 scan idiom:
// row_shr 1/2/4/8 inside each 16-lane row, row_bcast15 / row_bcast31 across rows, total in lane 63.
#define SCV_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (rmask), 0xf, false)

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    uint32_t t;
    t = (uint32_t)SCV_DPP(0, v, 0x111, 0xf); v = t > v ? t : v;   // row_shr:1
    t = (uint32_t)SCV_DPP(0, v, 0x112, 0xf); v = t > v ? t : v;   // row_shr:2
    t = (uint32_t)SCV_DPP(0, v, 0x114, 0xf); v = t > v ? t : v;   // row_shr:4
    t = (uint32_t)SCV_DPP(0, v, 0x118, 0xf); v = t > v ? t : v;   // row_shr:8
    t = (uint32_t)SCV_DPP(0, v, 0x142, 0xa); v = t > v ? t : v;   // row_bcast:15 -> rows 1, 3
    t = (uint32_t)SCV_DPP(0, v, 0x143, 0xc); v = t > v ? t : v;   // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    uint32_t t;
    t = (uint32_t)SCV_DPP(-1, v, 0x111, 0xf); v = t < v ? t : v;
    t = (uint32_t)SCV_DPP(-1, v, 0x112, 0xf); v = t < v ? t : v;
    t = (uint32_t)SCV_DPP(-1, v, 0x114, 0xf); v = t < v ? t : v;
    t = (uint32_t)SCV_DPP(-1, v, 0x118, 0xf); v = t < v ? t : v;
    t = (uint32_t)SCV_DPP(-1, v, 0x142, 0xa); v = t < v ? t : v;
    t = (uint32_t)SCV_DPP(-1, v, 0x143, 0xc); v = t < v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)SCV_DPP(0, v, 0x111, 0xf);
    v += (uint32_t)SCV_DPP(0, v, 0x112, 0xf);
    v += (uint32_t)SCV_DPP(0, v, 0x114, 0xf);
    v += (uint32_t)SCV_DPP(0, v, 0x118, 0xf);
    v += (uint32_t)SCV_DPP(0, v, 0x142, 0xa);
    v += (uint32_t)SCV_DPP(0, v, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit sum as three 32-bit lane sums of 22/21/21-bit limbs (each limb sum < 64 * 2^22 = 2^28): exact
// for any int64 inputs modulo 2^64, no carries across lanes needed.
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    const unsigned long long u = (unsigned long long)v;
    const unsigned long long s0 = wave_sum_u32((uint32_t)(u & 0x3fffffu));
    const unsigned long long s1 = wave_sum_u32((uint32_t)((u >> 22) & 0x1fffffu));
    const unsigned long long s2 = wave_sum_u32((uint32_t)((u >> 43) & 0x1fffffu));
    return (long long)(s0 + (s1 << 22) + (s2 << 43));
}

// Agent-scope relaxed accesses (global_store / global_load ... sc1): stores are written through, loads bypass the
// CU's L1, so data handed from one workgroup to another INSIDE a launch needs no release / acquire fence -- only
// that every storing wave drains its stores (s_waitcnt vmcnt(0)) before the arrival counter is bumped
// (guide: cdna_hip_programming.md Guideline 16, recipe R1).
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

typedef unsigned short scv_v2h __attribute__((ext_vector_type(2)));

template <int RL2>
__device__ __forceinline__ void vote(uint32_t* hist, uint32_t copy, uint32_t v, uint32_t& bad) {
    bad |= v;
    const uint32_t bin = v < 1023u ? v : 1023u;  // out-of-domain -> bin 1023 (+ error flag)
    atomicAdd(&hist[(bin << RL2) | copy], 1u);   // ds_add_u32, no return
}

template <int RL2>
__device__ __forceinline__ void vote4(uint32_t* hist, uint32_t copy, const int4& x, uint32_t& bad) {
    vote<RL2>(hist, copy, (uint32_t)x.x, bad);
    vote<RL2>(hist, copy, (uint32_t)x.y, bad);
    vote<RL2>(hist, copy, (uint32_t)x.z, bad);
    vote<RL2>(hist, copy, (uint32_t)x.w, bad);
}

typedef int v4i32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int4 stream_load(const int4* p) {
    // read-once stream: non-temporal so the line is not kept for a reuse that never comes
    const v4i32 v = __builtin_nontemporal_load(reinterpret_cast<const v4i32*>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t scv_v2u __attribute__((ext_vector_type(2)));
typedef uint32_t scv_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) scv_v2u lds_v2u;
typedef __attribute__((address_space(3))) scv_v4u lds_v4u;

// One LDS-DMA piece: 64 lanes x 16 bytes, lane l's bytes (from gbase + goff_l) land at lds_dst + 16 l.  No VGPR is staged: the bytes
// in flight cost no registers.  The source is a scalar base + a 32-bit lane offset; M0 (the LDS destination) is written in the
// statement that reads it and restored; non-temporal (a read-once stream).  Counted by vmcnt -- which hipcc does not know about.
__device__ __forceinline__ void lds_dma16(const void* gbase, uint32_t goff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(goff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int64_t uniform64(int64_t v) {      // tell the compiler a value is wave-uniform (it lives in SGPRs from here on)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v & 0xffffffffu));
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Stream the 16-byte vectors [lo, hi) of one cell into the replicated LDS histogram:
// U loads in flight per lane, each wave instruction covering 1 KiB contiguous.
template <int RL2, int T, int U>
__device__ __forceinline__ void stream_votes(uint32_t* hist, uint32_t copy, const int4* v4, int64_t lo, int64_t hi,
                                             int tid, uint32_t& bad) {
    int64_t i = lo + tid;
    for (; i + (int64_t)(U - 1) * T < hi; i += (int64_t)U * T) {
        int4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = stream_load(v4 + i + (int64_t)u * T);
#pragma unroll
        for (int u = 0; u < U; ++u) vote4<RL2>(hist, copy, x[u], bad);
    }
    for (; i < hi; i += T) {
        const int4 x = stream_load(v4 + i);
        vote4<RL2>(hist, copy, x, bad);
    }
}

// ---- work-item traversal ---------------------------------------------------------------------
//
// A work item is a (cell, segment).  Cells are traversed budget-major in DESCENDING n_valid order
// (`ord`, built in LDS by every workgroup from n_valid[B]): with ragged prefix budgets
// (o1.py:274-276, n_valid = 1,1,...,2,4,8,...) a problem-major static stride would hand one
// workgroup all the long budgets; sorted budget-major striding gives every workgroup the same mix
// and puts the long cells first (LPT).  With equal n_valid it is just another order.

constexpr int kMaxSortedB = 512;

__device__ __forceinline__ int64_t valid_len(const AggArgs& a, int32_t b) {
    if (!a.n_valid) return a.N;
    const int64_t nv = a.n_valid[b];
    return nv < 0 ? 0 : (nv > a.N ? a.N : nv);
}

// Rank of budget b (nb votes) among the launch's budgets by valid length, ties by index: the number of budgets in front of it.  Every
// lane of the wave calls it together (lanes without a budget pass any b): the lengths arrive 64 at a time with ONE coalesced load and
// are handed round with v_readlane -- round 5: the obvious loop (a load of n_valid[c] per comparison, each waiting for the one before)
// cost 0.65 us per comparison, B^2 of them: 32 us of a 57 us launch at 7 budgets, 53 us at 9.
template <bool DESCENDING, typename F>
__device__ __forceinline__ int budget_rank_of(const AggArgs& a, int32_t b, int64_t nb, F key_of) {
    const int lane = (int)threadIdx.x & 63;
    int rank = 0;
    for (int c0 = 0; c0 < a.B; c0 += 64) {
        const int cc = c0 + lane;
        const int32_t mine = cc < a.B ? (int32_t)key_of(cc) : 0;
        const int m = a.B - c0 < 64 ? a.B - c0 : 64;
        for (int j = 0; j < m; ++j) {
            const int64_t nc = (int64_t)__builtin_amdgcn_readlane(mine, j);
            const int c = c0 + j;
            rank += DESCENDING ? ((nc > nb) || (nc == nb && c < b)) : ((nc < nb) || (nc == nb && c < b));
        }
    }
    return rank;
}
template <bool DESCENDING>
__device__ __forceinline__ int budget_rank(const AggArgs& a, int32_t b, int64_t nb) {
    return budget_rank_of<DESCENDING>(a, b, nb, [&](int c) { return valid_len(a, c); });
}

// ---- scv_sort_prefix (scvote_sort_prefix.hip.h): which budget lists it serves ---------------------------------------------------
constexpr int sv_log2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
// class of a budget of n votes over pool rows of N votes on the shape of NV votes per lane (NV / 2 < N <= NV): 0 = no votes |
// 1 + j = the first 2^j votes, 2^j <= NV / 2 | log2(NV / 2) + 2 = all N votes | -1 = not served
template <int NV>
__device__ __forceinline__ int sort_prefix_class(int64_t n, int64_t N) {
    constexpr int NP = NV / 2;
    if (n <= 0) return 0;
    if (n >= N) return sv_log2(NP) + 2;
    if ((n & (n - 1)) == 0 && n <= NP) return 1 + (31 - __builtin_clz((uint32_t)n));
    return -1;
}
// true when every budget of the launch has a class (every thread of the workgroup calls it: one __syncthreads inside)
template <int NV>
__device__ __forceinline__ bool sort_prefix_serves(const AggArgs& a, int tid, int nthreads) {
    int bad = 0;
    for (int b = tid; b < a.B; b += nthreads) bad |= sort_prefix_class<NV>(valid_len(a, b), a.N) < 0 ? 1 : 0;
    return __syncthreads_or(bad) == 0;
}
// ... asked by the kernels the host queues behind scv_sort_prefix (a.skip_sortable = its NV; 0: never)
__device__ __forceinline__ bool sort_prefix_took_it(const AggArgs& a, int tid, int nthreads) {
    if (a.skip_sortable == 128) return sort_prefix_serves<128>(a, tid, nthreads);
    if (a.skip_sortable == 64) return sort_prefix_serves<64>(a, tid, nthreads);
    if (a.skip_sortable == 32) return sort_prefix_serves<32>(a, tid, nthreads);
    return false;
}

// returns true when `ord` is in use (caller must __syncthreads() before reading it)
__device__ __forceinline__ bool build_budget_order(const AggArgs& a, int32_t* ord, int tid, int nthreads) {
    if (!a.sorted || !a.n_valid || a.B > kMaxSortedB) return false;
    for (int b0 = 0; b0 < a.B; b0 += nthreads) {                     // (whole waves: budget_rank hands the lengths round the lanes)
        const int b = b0 + tid;
        const bool have = b < a.B;
        const int64_t nb = have ? valid_len(a, b) : 0;
        const int rank = budget_rank<true>(a, b, nb);
        if (have) ord[rank] = b;
    }
    return true;
}

__device__ __forceinline__ void item_to_cell(const AggArgs& a, bool use_ord, const int32_t* ord, int64_t ci,
                                             int64_t& p, int32_t& b) {
    if (use_ord) {
        const int64_t bi = ci / a.P;
        p = ci - bi * a.P;
        b = ord[bi];
    } else {
        p = ci / a.B;
        b = (int32_t)(ci - p * a.B);
    }
}

// Walks the traversal index ci -> (p, b) for a fixed stride without a 64-bit division per cell
// (item_to_cell costs ~130 instructions): one division at construction, then add-and-carry.
struct CellWalker {
    int64_t major, minor;     // natural: (p, b) = (major, minor), minor < B;  sorted: (bi, p) = (major, minor), minor < P
    int64_t dmajor, dminor, modulus;
    __device__ __forceinline__ CellWalker(const AggArgs& a, bool use_ord, int64_t ci0, int64_t stride) {
        modulus = use_ord ? a.P : (int64_t)a.B;
        major = ci0 / modulus; minor = ci0 - major * modulus;
        dmajor = stride / modulus; dminor = stride - dmajor * modulus;
    }
    __device__ __forceinline__ void advance() {
        major += dmajor; minor += dminor;
        if (minor >= modulus) { minor -= modulus; major += 1; }
    }
    __device__ __forceinline__ void get(bool use_ord, const int32_t* ord, int64_t& p, int32_t& b) const {
        if (use_ord) { p = minor; b = ord[major]; } else { p = major; b = (int32_t)minor; }
    }
};

// ---- cell epilogue shared by the streaming kernel and the split-N merge kernel ------------------
//
// statistics.multimode (statistics.py:599-601) + o1.py:204-213 on per-thread bin counts cnt[k]
// (bin = tid + k*T).  Precondition: red[48] was zeroed by thread 0 before the last barrier.
template <int T, bool TOK, bool XTRA = false>
__device__ __forceinline__ void finalize_cell(const AggArgs& a, uint32_t* red, const uint32_t (&cnt)[kBins / T],
                                              long long tsum, int tid, int64_t cell, int32_t b, int32_t truth) {
    constexpr int NB = kBins / T;
    constexpr int NW = T / 64;
    const int lane = tid & 63, wid = tid >> 6;
    uint32_t lmax = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        lmax = cnt[k] > lmax ? cnt[k] : lmax;
        if (tid + k * T == truth) red[48] = cnt[k];   // truth_count = histogram[truth] (pass@k's c)
    }
    const uint32_t wmax = wave_max_u32(lmax);
    if (lane == 0) red[wid] = wmax;
    if (TOK) {
        const long long wt = wave_sum_i64(tsum);
        if (lane == 0) {
            red[64 + 2 * wid] = (uint32_t)((unsigned long long)wt & 0xffffffffull);
            red[65 + 2 * wid] = (uint32_t)((unsigned long long)wt >> 32);
        }
    }
    __syncthreads();  // B2: per-wave maxima visible (and, in the streaming kernel, the histogram is zero again)

    uint32_t maxc = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const uint32_t m = red[w]; maxc = m > maxc ? m : maxc; }
    uint32_t nm = 0, mm = 1024u;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const uint32_t bin = (uint32_t)(tid + k * T);
        if (cnt[k] == maxc) { nm += 1; mm = bin < mm ? bin : mm; }
    }
    nm = wave_sum_u32(nm);
    mm = wave_min_u32(mm);
    if (lane == 0) { red[16 + wid] = nm; red[32 + wid] = mm; }
    __syncthreads();  // B3

    if (tid == 0) {
        uint32_t n_modes = 0, min_mode = 1024u;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            n_modes += red[16 + w];
            const uint32_t m = red[32 + w];
            min_mode = m < min_mode ? m : min_mode;
        }
        const uint32_t tc = red[48];
        long long tok = 0;
        if (TOK) {
#pragma unroll
            for (int w = 0; w < NW; ++w)
                tok += (long long)(((unsigned long long)red[65 + 2 * w] << 32) | red[64 + 2 * w]);
        }
        // o1.py:204-213: hit = truth in modes; multimode([]) == [] -> no hit when max_count == 0
        const bool any = maxc > 0;
        const uint32_t hit = (any && tc == maxc) ? 1u : 0u;
        if (!any) n_modes = 0;
        if (a.cells) {
            uint4 rec;
            rec.x = maxc;
            rec.y = tc;
            rec.z = (n_modes & 0xffffu) | ((any ? (min_mode & 0xffffu) : 0xffffu) << 16);
            rec.w = hit;
            if (XTRA && (a.overwrite | a.boot)) {      // read by other workgroups of this launch: write through
                unsigned long long* c8 = reinterpret_cast<unsigned long long*>(a.cells) + 2 * cell;
                st_agent(c8, (unsigned long long)rec.x | ((unsigned long long)rec.y << 32));
                st_agent(c8 + 1, (unsigned long long)rec.z | ((unsigned long long)rec.w << 32));
            } else reinterpret_cast<uint4*>(a.cells)[cell] = rec;
        }
        if (a.cell_tokens) {
            if (XTRA && (a.overwrite | a.boot)) st_agent(reinterpret_cast<unsigned long long*>(a.cell_tokens) + cell, (unsigned long long)tok);
            else a.cell_tokens[cell] = tok;
        }
        // o1.py:238-240 as integers: tie-class counter, token sum, truth-count sum
        if (a.tie_hits && hit) atomicAdd(&a.tie_hits[(int64_t)b * SCV_TIE_CLASSES + n_modes], 1ull);
        if (TOK && a.token_sum) atomicAdd(&a.token_sum[b], (unsigned long long)tok);
        if (a.truth_sum) atomicAdd(&a.truth_sum[b], (unsigned long long)tc);
    }
}

// votes + tokens, 16-byte vectors [lo, hi) of a run whose token row is 16-byte congruent with its vote row: UT loads of each
// stream in flight per lane
template <int RL2, int T, int UT>
__device__ __forceinline__ void stream_votes_tokens(uint32_t* hist, uint32_t copy, const int4* v4, const int4* t4, int64_t lo, int64_t hi,
                                                    int tid, uint32_t& bad, long long& tsum) {
    int64_t i = lo + tid;
    for (; i + (int64_t)(UT - 1) * T < hi; i += (int64_t)UT * T) {
        int4 x[UT], y[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            x[u] = stream_load(v4 + i + (int64_t)u * T);
            y[u] = stream_load(t4 + i + (int64_t)u * T);
        }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            vote4<RL2>(hist, copy, x[u], bad);
            tsum += (long long)y[u].x + (long long)y[u].y + (long long)y[u].z + (long long)y[u].w;
        }
    }
    for (; i < hi; i += T) {
        const int4 x = stream_load(v4 + i);
        const int4 y = stream_load(t4 + i);
        vote4<RL2>(hist, copy, x, bad);
        tsum += (long long)y.x + (long long)y.y + (long long)y.z + (long long)y.w;
    }
}

// (Round 4, measured and not kept: the TOKEN row moved by LDS-DMA -- global_load_lds_dwordx4, one U KiB staging block per wave, no
//  VGPR staged -- so that the vote stream keeps its U register loads per lane: twice the bytes in flight per lane.  On the same box,
//  interleaved with the round-3 library: 6.81 / 6.79 against 6.75 / 6.79 TB/s on 4 MiB rows, and C2 with tokens 47.8 against 45.0 us --
//  the register form with its cross-item prefetch is as fast on long rows and faster on short ones: profiles/r04_tokens_dma_ab.log.)
// Stream one contiguous run of votes (a whole cell, a split-N segment, or the run between two
// prefix boundaries) into the replicated histogram; accumulates the token sum when TOK.
template <int RL2, int T, int U, bool TOK>
__device__ __forceinline__ void stream_row(const AggArgs& a, uint32_t* hist, uint32_t copy, const int32_t* row,
                                           const int32_t* trow, int64_t n, int tid, uint32_t& bad, long long& tsum) {
    // head: scalars up to the first 16-byte boundary (rows are unaligned when N % 4 != 0)
    int64_t head = (int64_t)(((16u - (uint32_t)((uintptr_t)row & 15u)) & 15u) >> 2);
    if (head > n) head = n;
    if (tid < head) {
        vote<RL2>(hist, copy, (uint32_t)row[tid], bad);
        if (TOK) tsum += trow[tid];
    }
    const int4* v4 = reinterpret_cast<const int4*>(row + head);
    const int64_t nvec = (n - head) >> 2;
    int64_t i = tid;
    if (!TOK) {
        stream_votes<RL2, T, U>(hist, copy, v4, 0, nvec, tid, bad);
    } else {
        // the token row is 16-byte congruent with the vote row for the layouts the ABI accepts when
        // both bases are; a token base that is not takes the scalar route.
        const bool tok_vec = (((uintptr_t)(trow + head)) & 15u) == 0;
        if (tok_vec) {
            const int4* t4 = reinterpret_cast<const int4*>(trow + head);
            constexpr int UT = U > 1 ? U / 2 : 1;
            for (; i + (int64_t)(UT - 1) * T < nvec; i += (int64_t)UT * T) {
                int4 x[UT], y[UT];
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    x[u] = stream_load(v4 + i + (int64_t)u * T);
                    y[u] = stream_load(t4 + i + (int64_t)u * T);
                }
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    vote4<RL2>(hist, copy, x[u], bad);
                    tsum += (long long)y[u].x + (long long)y[u].y + (long long)y[u].z + (long long)y[u].w;
                }
            }
            for (; i < nvec; i += T) {
                const int4 x = stream_load(v4 + i);
                const int4 y = stream_load(t4 + i);
                vote4<RL2>(hist, copy, x, bad);
                tsum += (long long)y.x + (long long)y.y + (long long)y.z + (long long)y.w;
            }
        } else {
            for (; i < nvec; i += T) {
                const int4 x = stream_load(v4 + i);
                vote4<RL2>(hist, copy, x, bad);
                const int32_t* ts = trow + head + 4 * i;
                tsum += (long long)ts[0] + (long long)ts[1] + (long long)ts[2] + (long long)ts[3];
            }
        }
    }
    {   // tail: the < 4 votes after the last full 16-byte vector
        const int64_t t0 = head + (nvec << 2);
        if (tid < n - t0) {
            vote<RL2>(hist, copy, (uint32_t)row[t0 + tid], bad);
            if (TOK) tsum += trow[t0 + tid];
        }
    }
}

// Fold the R copies of every bin owned by this thread (bin = tid + k*T); ZERO re-arms the histogram.
template <int RL2, int T, bool ZERO>
__device__ __forceinline__ void fold_copies(uint32_t* hist, int tid, uint32_t (&cnt)[kBins / T]) {
    constexpr int R = 1 << RL2, CH = R / 4, BPR = 64 / R;
#pragma unroll
    for (int k = 0; k < kBins / T; ++k) {
        const int bin = tid + k * T;
        uint4* h4 = reinterpret_cast<uint4*>(hist + (bin << RL2));
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int jj = (j + bin / BPR) & (CH - 1);   // rotate: the 16-lane groups of a b128 read hit 16 distinct slots
            const uint4 x = h4[jj];
            s += x.x + x.y + x.z + x.w;
            if (ZERO) h4[jj] = make_uint4(0, 0, 0, 0);
        }
        cnt[k] = s;
    }
}

// One work item of the streaming kernel, resolved to pointers: (cell, segment) -> votes [row, row + n).
struct StreamItem {
    const int32_t* row;
    const int4* v4;       // first 16-byte aligned vector of the run
    int64_t n, head, nvec, cell, p;
    int32_t b;
};

__device__ __forceinline__ void describe_item(const AggArgs& a, bool use_ord, const int32_t* ord, int64_t item,
                                              StreamItem& it, int64_t& lo) {
    const int32_t S = a.segs;
    const int64_t ci = S > 1 ? item / S : item;
    const int32_t seg = S > 1 ? (int32_t)(item - ci * S) : 0;
    item_to_cell(a, use_ord, ord, ci, it.p, it.b);
    it.cell = it.p * a.B + it.b;
    int64_t n = valid_len(a, it.b);
    lo = 0;
    if (S > 1) {                      // split-N: this workgroup owns votes [lo, lo + n) of the cell
        lo = (int64_t)seg * a.seg_len;
        int64_t hi = lo + a.seg_len;
        hi = hi > n ? n : hi;
        n = hi > lo ? hi - lo : 0;
    }
    it.n = n;
    it.row = a.answers + it.cell * a.N + lo;
    int64_t head = (int64_t)(((16u - (uint32_t)((uintptr_t)it.row & 15u)) & 15u) >> 2);
    it.head = head > n ? n : head;
    it.v4 = reinterpret_cast<const int4*>(it.row + it.head);
    it.nvec = (n - it.head) >> 2;
}

// splitmix64 finaliser and friends (spec: include/scvote.h); used by the generator and by both bootstrap kernels
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ uint32_t mulhi32(uint32_t x, uint32_t n) { return __umulhi(x, n); }
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

// ---- single-launch epilogues of the streaming kernel ------------------------------------------------
//
// (a) counters without a memset or a reduce launch: every workgroup bumps tickets[0] when it has finished its
//     items; the one that finds gridDim.x - 1 there knows every cell record of the launch has been written (write-
//     through) and OVERWRITES the per-budget counters from the cell table (o1.py:236-245 as integers).
template <int T, bool TOK>
__device__ __forceinline__ void overwrite_counters_from_cells(const AggArgs& a, uint32_t* lds, int lds_words /* >= 4096 */, int tid) {
    // budgets are handled GB at a time, all in parallel: LDS holds GB tie-class rows of 1025 words + 2 * GB 64-bit sums
    const unsigned long long* c8 = reinterpret_cast<const unsigned long long*>(a.cells);
    const int32_t fit = (lds_words - 16) / (SCV_TIE_CLASSES + 4);           // 3 at the smallest histogram (R = 4), 15 at R = 16
    const int32_t GB = a.B < fit ? a.B : fit;
    uint32_t* tie = lds;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds + ((GB * SCV_TIE_CLASSES + 1) & ~1));   // [GB] truth sums, [GB] token sums
    for (int32_t b0 = 0; b0 < a.B; b0 += GB) {
        const int32_t nb = a.B - b0 < GB ? a.B - b0 : GB;
        for (int i = tid; i < nb * SCV_TIE_CLASSES; i += T) tie[i] = 0;
        if (tid < 2 * GB) acc[tid] = 0;
        __syncthreads();
        // thread -> (problem, budget of the group): consecutive threads read consecutive cells of one problem row
        const int64_t work = a.P * nb;
        for (int64_t w = tid; w < work; w += T) {
            const int64_t p = w / nb;
            const int32_t bl = (int32_t)(w - p * nb);
            const int64_t cell = p * a.B + b0 + bl;
            const unsigned long long lo = ld_agent(c8 + 2 * cell), hi = ld_agent(c8 + 2 * cell + 1);
            if ((hi >> 32) & 0xffu) atomicAdd(&tie[bl * SCV_TIE_CLASSES + ((uint32_t)hi & 0xffffu)], 1u);
            atomicAdd(&acc[bl], lo >> 32);
            if (TOK && a.ow_tok) atomicAdd(&acc[GB + bl], ld_agent(reinterpret_cast<const unsigned long long*>(a.cell_tokens) + cell));
        }
        __syncthreads();
        if (a.ow_tie)
            for (int i = tid; i < nb * SCV_TIE_CLASSES; i += T) a.ow_tie[(int64_t)b0 * SCV_TIE_CLASSES + i] = tie[i];
        if (tid < nb) {
            if (a.ow_truth) a.ow_truth[b0 + tid] = acc[tid];
            if (a.ow_tok) a.ow_tok[b0 + tid] = TOK ? acc[GB + tid] : 0ull;
        }
        __syncthreads();
    }
}

// (c) Bootstrap in the SAME launch as the vote (north_star: "fused in the same launch").  The problem-
//     level bootstrap needs every cell of the launch, so all workgroups meet at a grid barrier after their last
//     cell (one arrival counter + a generation word).  Co-residency is the runtime's promise: the host launches this
//     form with hipLaunchCooperativeKernel (grid <= the occupancy query; refused launches fall back to two kernels).
//     Belt and braces: the spin is bounded (a.boot_spins polls); a workgroup that gives up raises error bit 4, skips
//     its resamples, and the host re-runs the WHOLE bootstrap as a separate launch at the next scv_sync and clears the
//     bit (csrc/scvote.hip: recover_fused_bootstrap) -- a valid call never fails because of co-tenancy.  Then every workgroup
//     stages the write-through cell table as 2-byte codes in the LDS that held its histogram and runs its share of
//     the resamples exactly as scv_bootstrap_lds_k does.
template <int T>
__device__ __forceinline__ void bootstrap_in_launch(const AggArgs& a, uint32_t* lds, int tid, bool& overflow) {
    const int lane = tid & 63;
    const int64_t BM = (int64_t)a.B * a.boot_M;
    uint32_t* cnt = lds;
    uint16_t* tab = reinterpret_cast<uint16_t*>(lds + ((BM + 3) & ~(int64_t)3));
    const unsigned long long* c8 = reinterpret_cast<const unsigned long long*>(a.cells);
    for (int64_t i = tid; i < a.ncells; i += T) {
        const unsigned long long hi = ld_agent(c8 + 2 * i + 1);            // n_modes | min_mode << 16 | hit << 32
        tab[i] = ((hi >> 32) & 0xffu) ? (uint16_t)(hi & 0xffffu) : (uint16_t)0;
    }
    const int32_t M = a.boot_M;
    for (int32_t r = a.boot_r0 + (int32_t)blockIdx.x; r < a.boot_r1; r += (int32_t)gridDim.x) {
        for (int64_t i = tid; i < BM; i += T) cnt[i] = 0;
        __syncthreads();
        uint64_t arg = a.boot_seed + kGolden * ((uint64_t)r * (uint64_t)a.P + (uint64_t)tid + 1);
        const uint64_t darg = kGolden * (uint64_t)T;
        for (int64_t j0 = (int64_t)(tid - lane); j0 < a.P; j0 += T, arg += darg) {
            const int64_t j = j0 + lane;
            const bool live = j < a.P;
            const uint64_t u = mix64(arg);
            const int64_t idx = live ? (int64_t)mulhi32((uint32_t)(u >> 32), (uint32_t)a.P) : 0;
            for (int32_t b = 0; b < a.B; ++b) {
                const uint32_t code = live ? (uint32_t)tab[idx * a.B + b] : 0u;
                const uint64_t ones = __ballot(code == 1u);
                if (lane == 0 && ones && M > 1) atomicAdd(&cnt[(int64_t)b * M + 1], (uint32_t)__popcll(ones));
                if (code == 1u && M <= 1) overflow = true;
                if (code > 1u) {
                    if (code >= (uint32_t)M) overflow = true;
                    else atomicAdd(&cnt[(int64_t)b * M + code], 1u);
                }
            }
        }
        __syncthreads();
        unsigned long long* o = a.boot_out + (int64_t)(r - a.boot_r0) * BM;
        for (int64_t i = tid; i < BM; i += T) o[i] = cnt[i];
        __syncthreads();
    }
}

// ---- kernel 1: streaming histogram / argmax (large N) -------------------------------------------
// RL2 = log2(copies), T = threads per workgroup, U = 16-byte loads in flight per lane.
// XTRA: the single-launch epilogues (overwrite-counters, bootstrap behind a grid barrier) are compiled in.  They are
// a separate instantiation so that the default hot path keeps round 1's register allocation (with them in, the
// headline variant went from 0 to 36 bytes of scratch and from 18 to 41 spilled SGPRs).
template <int RL2, int T, int U, bool TOK, bool XTRA = false>
__global__ __launch_bounds__(T) void scv_hist_argmax(const AggArgs a) {
    constexpr int R = 1 << RL2;
    constexpr int NB = kBins / T;        // bins folded per thread in the epilogue
    static_assert(NB >= 1 && T / 64 <= 16, "workgroup shape");

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* hist = smem;
    uint32_t* red = smem + kBins * R;
    int32_t* ord = reinterpret_cast<int32_t*>(red + kRedWords);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t copy = (uint32_t)lane & (R - 1);

    {   // zero the replicated histogram once; afterwards the epilogue leaves it zeroed
        uint4* h4 = reinterpret_cast<uint4*>(hist);
        for (int i = tid; i < kBins * R / 4; i += T) h4[i] = make_uint4(0, 0, 0, 0);
    }
    const bool use_ord = build_budget_order(a, ord, tid, T);
    __syncthreads();

    const int32_t S = a.segs;
    const int64_t nitems = a.ncells * S;
    uint32_t bad = 0;
    // Cross-item prefetch (votes-only variant): the first U*T vectors of the NEXT item are loaded into
    // registers before the current item's epilogue (B1 / fold / reductions), so a short cell's load
    // latency overlaps the previous cell's epilogue instead of following it.
    const bool pf = !TOK && a.prefetch;
    // ... and with the tokens stream: the first UT vectors of BOTH rows of the next item (round 2 had no prefetch here: 6.4-6.6
    // against 6.9-7.1 TB/s votes-only).  Items whose token row is not 16-byte congruent with the vote row take stream_row.
    constexpr int UT = U > 1 ? U / 2 : 1;
    const bool pft = TOK && a.prefetch;
    int4 pret[TOK ? UT : 1];
    const int4* cur_t4 = nullptr;             // first aligned vector of the current item's token row (NULL: not congruent)
    auto token_vectors = [&](const StreamItem& it, int64_t lo) -> const int4* {
        const int32_t* t = a.tokens + it.cell * a.N + lo + it.head;
        return (((uintptr_t)t) & 15u) == 0 ? reinterpret_cast<const int4*>(t) : nullptr;
    };
    int4 pre[U];
    StreamItem cur;
    int64_t cur_lo = 0;
    if ((int64_t)blockIdx.x < nitems) {
        describe_item(a, use_ord, ord, blockIdx.x, cur, cur_lo);
        if (pf) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t idx = tid + (int64_t)u * T;
                pre[u] = idx < cur.nvec ? stream_load(cur.v4 + idx) : make_int4(0, 0, 0, 0);
            }
        }
        if (TOK && pft) {
            cur_t4 = token_vectors(cur, cur_lo);
            if (cur_t4) {
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    const int64_t idx = tid + (int64_t)u * T;
                    pre[u] = idx < cur.nvec ? stream_load(cur.v4 + idx) : make_int4(0, 0, 0, 0);
                    pret[u] = idx < cur.nvec ? stream_load(cur_t4 + idx) : make_int4(0, 0, 0, 0);
                }
            }
        }
    }
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int64_t cell = cur.cell, p = cur.p;
        const int32_t b = cur.b;
        long long tsum = 0;
        if (pf) {
            // o1.py:181-195 with the first tile already in registers
            if (tid < cur.head) vote<RL2>(hist, copy, (uint32_t)cur.row[tid], bad);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (tid + (int64_t)u * T < cur.nvec) vote4<RL2>(hist, copy, pre[u], bad);
            if (cur.nvec > (int64_t)U * T) stream_votes<RL2, T, U>(hist, copy, cur.v4, (int64_t)U * T, cur.nvec, tid, bad);
            const int64_t t0 = cur.head + (cur.nvec << 2);
            if (tid < cur.n - t0) vote<RL2>(hist, copy, (uint32_t)cur.row[t0 + tid], bad);
        } else if (TOK && pft && cur_t4) {
            // votes + tokens with the first tile of both rows already in registers
            const int32_t* trow = a.tokens + cell * a.N + cur_lo;
            if (tid < cur.head) { vote<RL2>(hist, copy, (uint32_t)cur.row[tid], bad); tsum += trow[tid]; }
#pragma unroll
            for (int u = 0; u < UT; ++u)
                if (tid + (int64_t)u * T < cur.nvec) {
                    vote4<RL2>(hist, copy, pre[u], bad);
                    tsum += (long long)pret[u].x + (long long)pret[u].y + (long long)pret[u].z + (long long)pret[u].w;
                }
            if (cur.nvec > (int64_t)UT * T) stream_votes_tokens<RL2, T, UT>(hist, copy, cur.v4, cur_t4, (int64_t)UT * T, cur.nvec, tid, bad, tsum);
            const int64_t t0 = cur.head + (cur.nvec << 2);
            if (tid < cur.n - t0) { vote<RL2>(hist, copy, (uint32_t)cur.row[t0 + tid], bad); tsum += trow[t0 + tid]; }
        } else {
            const int32_t* trow = TOK ? a.tokens + cell * a.N + cur_lo : nullptr;
            stream_row<RL2, T, U, TOK>(a, hist, copy, cur.row, trow, cur.n, tid, bad, tsum);   // o1.py:181-195
        }
        const bool more = item + gridDim.x < nitems;
        StreamItem nxt = cur;
        int64_t nxt_lo = 0;
        const int4* nxt_t4 = nullptr;
        if (more) {
            describe_item(a, use_ord, ord, item + gridDim.x, nxt, nxt_lo);
            if (pf) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t idx = tid + (int64_t)u * T;
                    pre[u] = idx < nxt.nvec ? stream_load(nxt.v4 + idx) : make_int4(0, 0, 0, 0);
                }
            }
            if (TOK && pft) {
                nxt_t4 = token_vectors(nxt, nxt_lo);
                if (nxt_t4) {
#pragma unroll
                    for (int u = 0; u < UT; ++u) {
                        const int64_t idx = tid + (int64_t)u * T;
                        pre[u] = idx < nxt.nvec ? stream_load(nxt.v4 + idx) : make_int4(0, 0, 0, 0);
                        pret[u] = idx < nxt.nvec ? stream_load(nxt_t4 + idx) : make_int4(0, 0, 0, 0);
                    }
                }
            }
        }
        if (tid == 0) red[48] = 0;
        __syncthreads();  // B1: all votes of this item are in LDS

        uint32_t cnt[NB];
        fold_copies<RL2, T, true>(hist, tid, cnt);     // and zero them for the next item
        if (S > 1) {
            // split-N, ONE launch (round 6; round 5 published 4 KiB per segment and a second kernel merged them: 27.8 us for one cell of
            // 2^24 votes): the segment's folded counts are ADDED into the cell's histogram in memory -- device atomics at L2, a wave's 64
            // consecutive bins = two 128-byte requests --, the token sum into the cell's 64-bit word; the workgroup that arrives last at the
            // cell's ticket reads the sums back (agent-scope loads), clears histogram, token word and ticket for the next launch, and runs the
            // common epilogue.  Nothing but the votes is read twice; the scratch is all-zero whenever no launch is in flight.
            uint32_t* gh = a.partial + (cell << 10);            // (split items are numbered cell-major and the sorted traversal is off: item / S == cell)
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (cnt[k]) __hip_atomic_fetch_add(gh + tid + k * T, cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (TOK) {
                const long long wt = wave_sum_i64(tsum);
                if (lane == 0 && wt) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(a.partial_tok) + cell, (unsigned long long)wt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            drain_stores();                                   // this wave's atomics have been performed at L2 ...
            __syncthreads();                                  // ... and every wave's (and the histogram is zero again)
            if (tid == 0) {
                const uint32_t arrived = __hip_atomic_fetch_add(a.tickets + 8 + cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                red[52] = (arrived == (uint32_t)S - 1u) ? 1u : 0u;
                red[48] = 0;
            }
            __syncthreads();
            if (red[52]) {                                    // the last segment of the cell: every other segment's sums are in memory
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    cnt[k] = ld_agent(gh + tid + k * T);
                    st_agent(gh + tid + k * T, 0u);
                }
                long long tsum_all = 0;
                if (TOK && tid == 0) {
                    unsigned long long* tw = reinterpret_cast<unsigned long long*>(a.partial_tok) + cell;
                    tsum_all = (long long)ld_agent(tw);
                    st_agent(tw, 0ull);
                }
                if (tid == 0) st_agent(a.tickets + 8 + cell, 0u);
                finalize_cell<T, TOK, XTRA>(a, red, cnt, tsum_all, tid, cell, b, a.truth[p]);
            }
        } else {
            finalize_cell<T, TOK, XTRA>(a, red, cnt, tsum, tid, cell, b, a.truth[p]);
            // the next item's votes may start: the histogram was re-zeroed before B2, and `red` is
            // next written after the next B1, which thread 0 only reaches after finalize_cell.
        }
        cur = nxt;
        cur_lo = nxt_lo;
        cur_t4 = nxt_t4;
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    if (XTRA && a.overwrite) {
        // the last workgroup to finish turns the cell table into the per-budget counters (overwriting them)
        drain_stores();                               // thread 0's write-through cell records
        __syncthreads();
        if (tid == 0) {
            const uint32_t arrived = atomicAdd(a.tickets, 1u);
            red[50] = (arrived == gridDim.x - 1u) ? 1u : 0u;
            if (red[50]) st_agent(a.tickets, 0u);
        }
        __syncthreads();
        if (red[50]) overwrite_counters_from_cells<T, TOK>(a, hist, kBins * R, tid);
    }
    if (XTRA && a.boot) {
        drain_stores();                               // thread 0's write-through cell records
        __syncthreads();
        if (tid == 0) {
            const uint32_t g0 = ld_agent(a.tickets + 2);          // read BEFORE arriving: it can only change after we have
            const uint32_t arrived = atomicAdd(a.tickets + 1, 1u);
            uint32_t ok = 1;
            if (arrived == gridDim.x - 1u) {
                st_agent(a.tickets + 1, 0u);
                atomicAdd(a.tickets + 2, 1u);                     // release everybody
            } else {
                uint32_t spins = 0;
                while (ld_agent(a.tickets + 2) == g0) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > a.boot_spins) { ok = 0; break; }  // not co-resident after all: give up; scv_sync re-runs the bootstrap
                }
            }
            red[51] = ok;
            if (!ok) atomicOr(a.err_flag, 4u);
        }
        __syncthreads();
        bool overflow = false;
        if (red[51]) bootstrap_in_launch<T>(a, hist, tid, overflow);
        if (overflow) atomicOr(a.err_flag, 2u);
    }
}

// ---- kernel 1d: per-budget reduction of the cell table (o1.py:236-245 as integers) ----------------
// Used instead of per-cell global atomics when there are many cells: hundreds of thousands of
// same-address device atomics serialise at ~12 ns each and would dominate small-N workloads.
// grid = (chunks, B); each workgroup reduces a block of problems for one budget.
template <bool TOK>
__global__ __launch_bounds__(256) void scv_reduce_cells(const scv_cell* cells, const int64_t* cell_tokens, int64_t P,
                                                        int32_t B, unsigned long long* tie_hits,
                                                        unsigned long long* token_sum, unsigned long long* truth_sum) {
    __shared__ uint32_t tie[SCV_TIE_CLASSES];
    __shared__ unsigned long long acc[2];
    const int tid = threadIdx.x;
    const int64_t per = (P + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per;
    const int64_t p1 = p0 + per < P ? p0 + per : P;
    const uint4* c4 = reinterpret_cast<const uint4*>(cells);
    for (int32_t b = blockIdx.y; b < B; b += gridDim.y) {          // gridDim.y is capped at 65535
        for (int i = tid; i < SCV_TIE_CLASSES; i += 256) tie[i] = 0;
        if (tid < 2) acc[tid] = 0;
        __syncthreads();
        unsigned long long tcs = 0;
        long long tks = 0;
        for (int64_t p = p0 + tid; p < p1; p += 256) {
            const uint4 c = c4[p * B + b];
            if (c.w & 0xffu) atomicAdd(&tie[c.z & 0xffffu], 1u);
            tcs += c.y;
            if (TOK) tks += cell_tokens[p * B + b];
        }
        tcs = (unsigned long long)wave_sum_i64((long long)tcs);
        if (TOK) tks = wave_sum_i64(tks);
        if ((tid & 63) == 0) {
            atomicAdd(&acc[0], tcs);
            if (TOK) atomicAdd(&acc[1], (unsigned long long)tks);
        }
        __syncthreads();
        if (tie_hits)
            for (int i = tid; i < SCV_TIE_CLASSES; i += 256)
                if (tie[i]) atomicAdd(&tie_hits[(int64_t)b * SCV_TIE_CLASSES + i], (unsigned long long)tie[i]);
        if (tid == 0) {
            if (truth_sum && acc[0]) atomicAdd(&truth_sum[b], acc[0]);
            if (TOK && token_sum && acc[1]) atomicAdd(&token_sum[b], acc[1]);
        }
        __syncthreads();                                            // before the next budget re-zeroes tie / acc
    }
}

// votes_at_max is an exact multiple of maxc (every modal value is voted maxc times) and the quotient is
// <= 1024, so a float reciprocal multiply rounded to nearest is exact -- ~4 instructions instead of the
// ~25 of an emulated 32-bit integer division (these kernels are instruction-issue bound).
__device__ __forceinline__ uint32_t exact_quotient(uint32_t votes_at_max, uint32_t maxc) {
    return (uint32_t)((float)votes_at_max * __builtin_amdgcn_rcpf((float)maxc) + 0.5f);
}

// ---- DPP helpers of the cell kernels ---------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E;                      // quad_perm [1,0,3,2] [2,3,0,1]
constexpr int kHalfMirror = 0x141, kRowMirror = 0x140;                // lane i -> 7-i / 15-i inside its 8 / 16 lanes

// ---- kernel 1f': tiny cells, ONE LANE PER CELL (N <= 32; the reference's own N = 1, 2, 4, 8 .. 32) ----
//
// scv_tiny_cells spreads a cell over G lanes and pays ~10 (N = 8) to ~100 (N = 32) VALU wave-instructions per cell
// in cross-lane rotations and group reductions (PMC: VALU-bound, profiles/r02_regimes_pmc_baseline.md), then a
// second launch reduces the cell table into the counters.  Here every lane owns a whole cell in registers: it
// loads its NV votes (lane-contiguous rows), counts equal pairs without any cross-lane traffic (3 instructions
// per pair), and derives max count / #modes / min mode / truth count from its own registers: ~3 (NV = 8) to ~29
// (NV = 32) wave-instructions per cell.  Votes past the valid prefix become unique sentinels that match nothing.
// The per-budget counters are accumulated in LDS per workgroup (tie classes cannot exceed NV) and flushed with
// one global atomic per non-zero counter -- one launch, and the cell table is written only if the caller wants it.
template <int NV, int T, bool TOK>
__global__ __launch_bounds__(T) void scv_lane_cells(const AggArgs a) {
    constexpr int TC = NV + 1;                                       // tie classes 0..NV
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* tie = lds;                                             // [B][TC]
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds + (((int64_t)a.B * TC + 1) & ~(int64_t)1));   // [B] truth sums | [B] token sums
    const int tid = threadIdx.x;
    const bool counters = a.tie_hits || a.truth_sum || (TOK && a.token_sum);
    if (counters) {
        for (int64_t i = tid; i < (int64_t)a.B * TC; i += T) tie[i] = 0;
        for (int i = tid; i < 2 * a.B; i += T) acc[i] = 0;
        __syncthreads();
    }
    const bool vec = a.wave_lds_words != 0;                          // host: N % 4 == 0 and 16-byte aligned bases
    const int32_t N = (int32_t)a.N;
    const int64_t stride = (int64_t)gridDim.x * T;
    int64_t ncell = (int64_t)blockIdx.x * T + tid;                   // walker of the NEXT cell to load
    int64_t np = ncell / a.B;
    int32_t nb = (int32_t)(ncell - np * a.B);
    const int64_t dp = stride / a.B;
    const int32_t db = (int32_t)(stride - dp * a.B);
    // stride % B == 0 (the host rounds the grid): a lane sees ONE budget, so its class-1 hits and its sums stay in
    // registers until the end -- per-cell LDS atomics from 64 lanes on the same word are serialised 64 deep
    const bool fixed_b = counters && db == 0;
    const int32_t my_b = nb;
    uint32_t h1 = 0;
    unsigned long long tcs = 0;
    long long toks = 0;

    struct Cell {
        uint32_t x[NV];
        int32_t tk[TOK ? NV : 1];
        int64_t cell;
        int32_t b, truth;
        uint32_t n;
    };
    // unconditional loads (a lane past the last cell reads cell 0; elements past N are not loaded: N is uniform)
    auto load = [&](Cell& c) {
        c.cell = ncell; c.b = nb;
        const bool live = ncell < a.ncells;
        c.n = live ? (uint32_t)valid_len(a, live ? nb : 0) : 0u;
        c.truth = a.truth[live ? np : 0];
        const int64_t off = (live ? (a.pool_rows ? np : ncell) : 0) * a.N;
        const int32_t* row = a.answers + off;
        const int32_t* trow = TOK ? a.tokens + off : nullptr;
        if (vec) {
#pragma unroll
            for (int k = 0; k < NV / 4; ++k) {
                int4 q = make_int4(0, 0, 0, 0), y = make_int4(0, 0, 0, 0);
                if (4 * k < N) {
                    q = stream_load(reinterpret_cast<const int4*>(row) + k);
                    if (TOK) y = stream_load(reinterpret_cast<const int4*>(trow) + k);
                }
                c.x[4 * k] = (uint32_t)q.x; c.x[4 * k + 1] = (uint32_t)q.y; c.x[4 * k + 2] = (uint32_t)q.z; c.x[4 * k + 3] = (uint32_t)q.w;
                if (TOK) { c.tk[4 * k] = y.x; c.tk[4 * k + 1] = y.y; c.tk[4 * k + 2] = y.z; c.tk[4 * k + 3] = y.w; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                c.x[i] = 0;
                if (TOK) c.tk[i] = 0;
                if (i < N) {
                    c.x[i] = (uint32_t)__builtin_nontemporal_load(row + i);
                    if (TOK) c.tk[i] = __builtin_nontemporal_load(trow + i);
                }
            }
        }
        ncell += stride; np += dp; nb += db;
        if (nb >= a.B) { nb -= a.B; np += 1; }
    };

    uint32_t bad = 0;
    auto count = [&](const Cell& c) {
        const uint32_t n = c.n;
        uint32_t x[NV], cnt[NV];
        long long tok = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool on = (uint32_t)i < n;
            const uint32_t v = c.x[i];
            bad |= on ? v : 0u;
            x[i] = on ? (v < 1023u ? v : 1023u) : (0xffff0000u + (uint32_t)i);   // an inactive slot matches nothing
            cnt[i] = 1;
            if (TOK) tok += on ? (long long)c.tk[i] : 0ll;
        }
        uint32_t best = 0, at_max = 0, tc = 0, n_modes_raw;
        const uint32_t tcmp = (c.truth >= 0 && c.truth < kBins) ? (uint32_t)c.truth : 0xffffffffu;
        if constexpr (NV <= 8) {
            // o1.py:181-195 + statistics.py:599: count_i = #{ j : x_j == x_i }, every pair once (28 pairs at NV = 8)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
#pragma unroll
                for (int j = i + 1; j < NV; ++j) {
                    const uint32_t e = x[i] == x[j] ? 1u : 0u;
                    cnt[i] += e;
                    cnt[j] += e;
                }
            }
            // one max over (count << 10 | 1023 - bin): max_count and the smallest modal bin together
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const uint32_t key = (uint32_t)i < n ? ((cnt[i] << 10) | (1023u - x[i])) : 0u;
                cnt[i] = key;
                best = key > best ? key : best;
            }
            const uint32_t thr = best & ~1023u;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                at_max += cnt[i] >= thr ? 1u : 0u;                   // inactive keys are 0: they only count when n == 0
                tc += x[i] == tcmp ? 1u : 0u;
            }
            n_modes_raw = (best >> 10) ? exact_quotient(at_max, best >> 10) : 0u;   // votes at max / max = distinct modes
        } else {
            // 16 / 32 votes: the pair triangle (120 / 496 compares, each a live predicate) loses to a bitonic sorting
            // network on min / max (80 / 240 exchanges, no predicates) followed by a run-length scan of the sorted votes:
            // run_i = length of the run of equal values ending at i; a run of the maximal length is counted once, at its
            // last element, so len(multimode) needs no division.  Inactive sentinels sort to the end and count for nothing.
#pragma unroll
            for (int k = 2; k <= NV; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int l = i ^ j;
                        if (l > i) {
                            const uint32_t lo = x[i] < x[l] ? x[i] : x[l], hi = x[i] < x[l] ? x[l] : x[i];
                            if ((i & k) == 0) { x[i] = lo; x[l] = hi; } else { x[i] = hi; x[l] = lo; }
                        }
                    }
                }
            }
            uint32_t run = 0, prev = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                run = x[i] == prev ? run + 1u : 1u;
                prev = x[i];
                const uint32_t key = x[i] < 0xffff0000u ? ((run << 10) | (1023u - x[i])) : 0u;
                cnt[i] = key;
                best = key > best ? key : best;
                tc += x[i] == tcmp ? 1u : 0u;
            }
            const uint32_t thr = best & ~1023u;
#pragma unroll
            for (int i = 0; i < NV; ++i) at_max += (cnt[i] >= thr && cnt[i]) ? 1u : 0u;   // ends of maximal runs
            n_modes_raw = at_max;
        }
        const uint32_t maxc = best >> 10;
        if (c.cell < a.ncells) {
            const bool any = maxc > 0;
            const uint32_t n_modes = any ? n_modes_raw : 0u;
            const uint32_t mm = 1023u - (best & 1023u);
            const uint32_t hit = (any && tc == maxc) ? 1u : 0u;                 // o1.py:206
            if (a.cells) {
                uint4 rec;
                rec.x = maxc;
                rec.y = tc;
                rec.z = (n_modes & 0xffffu) | ((any ? (mm & 0xffffu) : 0xffffu) << 16);
                rec.w = hit;
                if (a.packed_cells) reinterpret_cast<uint32_t*>(a.cells)[c.cell] = pack_cell(maxc, tc, n_modes, mm, hit);
                else reinterpret_cast<uint4*>(a.cells)[c.cell] = rec;
            }
            if (TOK && a.cell_tokens) a.cell_tokens[c.cell] = tok;
            if (fixed_b) {                                                        // o1.py:238-240 as integers
                h1 += (hit && n_modes == 1u) ? 1u : 0u;
                if (hit && n_modes != 1u) atomicAdd(&tie[c.b * TC + (int32_t)n_modes], 1u);
                tcs += tc;
                if (TOK) toks += tok;
            } else if (counters) {                                                // ... per workgroup in LDS
                if (hit) atomicAdd(&tie[c.b * TC + (int32_t)n_modes], 1u);
                if (tc) atomicAdd(&acc[c.b], (unsigned long long)tc);
                if (TOK) atomicAdd(&acc[a.B + c.b], (unsigned long long)tok);
            }
        }
    };

    // KC cells per lane and step: KC row loads are in flight behind the KC cells being counted.  With one cell per step a wave of
    // the N = 8 shape had 2 KiB in flight and ~6 dependent memory round trips per launch: latency, not the ~100 VALU instructions
    // per 64 cells, set its time (PMC, profiles/r03_lane_cells_pmc.md).
    constexpr int KC = NV <= 4 ? (TOK ? 2 : 4) : (NV == 8 && !TOK ? 2 : 1);      // (more would spill at the shapes' 128-VGPR bounds)
    Cell ca[KC], cb[KC];
    const int64_t last = a.ncells;                                   // lanes of a wave run the same number of steps
    const int64_t first = (int64_t)blockIdx.x * T + tid - (tid & 63);
    const int64_t kstride = (int64_t)KC * stride;                    // cells a step of the whole grid covers
#pragma unroll
    for (int k = 0; k < KC; ++k) load(ca[k]);
    for (int64_t c0 = first; c0 < last; c0 += 2 * kstride) {
        const bool more = c0 + kstride < last;
        if (more) {
#pragma unroll
            for (int k = 0; k < KC; ++k) load(cb[k]);
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) count(ca[k]);
        if (!more) break;
        if (c0 + 2 * kstride < last) {
#pragma unroll
            for (int k = 0; k < KC; ++k) load(ca[k]);
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) count(cb[k]);
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    if (fixed_b) {
        if (h1) atomicAdd(&tie[my_b * TC + 1], h1);
        if (tcs) atomicAdd(&acc[my_b], tcs);
        if (TOK && toks) atomicAdd(&acc[a.B + my_b], (unsigned long long)toks);
    }
    if (counters) {
        __syncthreads();
        for (int64_t i = tid; i < (int64_t)a.B * TC; i += T) {
            const uint32_t v = tie[i];
            if (v && a.tie_hits) {
                const int64_t b = i / TC;
                atomicAdd(&a.tie_hits[b * SCV_TIE_CLASSES + (i - b * TC)], (unsigned long long)v);
            }
        }
        for (int i = tid; i < a.B; i += T) {
            if (a.truth_sum && acc[i]) atomicAdd(&a.truth_sum[i], acc[i]);
            if (TOK && a.token_sum && acc[a.B + i]) atomicAdd(&a.token_sum[i], acc[a.B + i]);
        }
    }
}

// ---- kernel 1f0: cells of EXACTLY 1, 2 or 4 votes -- the reference's most common sizes ---------------------------------------
//
// o1.py:302 runs every ask-nicely budget with N = 1 (8 ... 20 budgets), o1.py:276 the first eight majority budgets with N = 1 and then
// N = 2, 4.  scv_lane_cells<4> serves them with its general machinery (a row per lane, a walker per cell, KC cells in flight):
// 157 lane-instructions per cell at N = 1 (profiles/r04_regimes.log: 1.0 TB/s of votes without the cell table).  A cell of N in
// {1, 2, 4} votes is 4 / 8 / 16 bytes: here a wave takes a BLOCK of 256 / 128 / 64 consecutive cells per step and lane l owns the cells
// block + l, block + 64 + l, ... (4 / N of them): every load and every store instruction of the wave touches CONSECUTIVE bytes (256 B
// ... 1 KiB of votes, 1 KiB of cell records) -- with 4 consecutive cells per lane the 16-byte records left in 64-byte strides and the
// launch took 1.6x the general kernel's time.  A cell is counted with a handful of compares (no sort, no sentinels: validity is a bit
// per vote).  With the grid rounded so that a step covers a multiple of B cells, a lane's cell slots keep their budgets for the whole
// launch: n_valid is read once and the class-1 hits and sums stay in registers.  Host contract: N == NV, 16-byte aligned bases,
// ncells % (256 / NV) == 0 (whole blocks), no pool rows, ncells < 2^29.
template <int NV, bool TOK>
__global__ __launch_bounds__(1024) void scv_few_votes(const AggArgs a) {
    static_assert(NV == 1 || NV == 2 || NV == 4, "cells of 1, 2 or 4 votes");
    constexpr int CPL = 4 / NV;                                      // cells per lane and step
    constexpr uint32_t BLK = 64u * CPL;                              // cells per wave and step
    constexpr int TC = NV + 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* tie = lds;                                             // [B][TC]
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds + (((int64_t)a.B * TC + 1) & ~(int64_t)1));   // [B] truth sums | [B] token sums
    const int tid = threadIdx.x, T = (int)blockDim.x, lane = tid & 63;
    const bool counters = a.tie_hits || a.truth_sum || (TOK && a.token_sum);
    if (counters) {
        for (int64_t i = tid; i < (int64_t)a.B * TC; i += T) tie[i] = 0;
        for (int i = tid; i < 2 * a.B; i += T) acc[i] = 0;
        __syncthreads();
    }
    const uint32_t B = (uint32_t)a.B;
    const uint32_t nblocks = (uint32_t)(a.ncells / BLK);
    const uint32_t nwaves = (uint32_t)gridDim.x * (uint32_t)(T >> 6);
    uint32_t blk = (uint32_t)blockIdx.x * (uint32_t)(T >> 6) + (uint32_t)(tid >> 6);          // this wave's block
    // (p, b) of each cell slot of the lane, then add-and-carry: a step moves every slot by nwaves * BLK cells = (dp, db)
    const uint32_t dcell = nwaves * BLK, dp = dcell / B, db = dcell - dp * B;
    const bool fixed_b = db == 0;                                    // the host rounds the grid: a lane's cells keep their budgets
    // Which cells of the block a lane owns.  With 16-byte records (the default cell table) lane l owns block + l, block + 64 + l ...: every record store
    // of the wave then touches consecutive bytes.  Round 6: with the 4-byte records of SCV_FLAG_PACKED_CELLS lane l owns the CPL consecutive cells
    // block + CPL l ...: its votes are ONE 16-byte load (N = 1: four cells, N = 2: two) instead of four / two 4- / 8-byte loads, its packed records one
    // 16- / 8-byte store (N = 1, 1.02e8 cells: 264 -> 241 us; 401 with 16-byte records).  Counters-only launches keep the first layout: they are bound by
    // their ~50 VALU per cell, not by the loads (the consecutive layout measured 131 -> 155 us there: more registers, fewer waves).
    const bool consec = NV < 4 && a.cells && a.packed_cells;
    auto slot_cell = [&](uint32_t bk, int j) -> uint64_t { return (uint64_t)bk * BLK + (consec ? (uint32_t)CPL * (uint32_t)lane + (uint32_t)j : 64u * (uint32_t)j + (uint32_t)lane); };
    uint32_t pj[CPL], bj[CPL], nj[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const uint32_t c = (uint32_t)slot_cell(blk, j);
        pj[j] = c / B;
        bj[j] = c - pj[j] * B;
        const int64_t n = valid_len(a, (int32_t)bj[j]);
        nj[j] = (uint32_t)(n < NV ? n : NV);
    }
    uint32_t h1[CPL];
    unsigned long long tcs[CPL];
    long long toks[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { h1[j] = 0; tcs[j] = 0; toks[j] = 0; }
    uint32_t bad = 0;
    typedef uint32_t vnv __attribute__((ext_vector_type(NV == 1 ? 1 : NV)));
    struct Votes { uint32_t w[CPL][NV]; int32_t tk[CPL][NV]; int32_t truth[CPL]; };
    // `ahead`: the slots' problems are those of the step after the current one (the slots' walkers plus one stride): the truth of a step
    // travels with its votes -- loaded inside the step, every step waited a memory latency for it (PMC round 4, N = 1: wave-wait 0.85;
    // counters only, N = 1: 183 -> 129 us, with tokens 217 -> 158; N = 2: 100 -> 97).  Not when the launch writes the cell table: that launch is bound by
    // its 16-byte records (5.3 TB/s of reads + writes at N = 1) and ran 7-9 % SLOWER with the extra loads in flight (r04_ab_few_truth.log).
    // Nor for cells of 4 votes (one truth per 16 bytes of votes: 91 -> 94 us with it).
    const bool pre = NV < 4 && !a.cells;
    auto load = [&](uint32_t bk, Votes& o, bool ahead) {            // every instruction: 64 lanes x 4 NV (consec: 16) consecutive bytes
        if constexpr (NV < 4) {
            if (consec) {
#pragma unroll
                for (int j = 0; j < CPL; ++j)
                    if (pre) o.truth[j] = a.truth[ahead ? pj[j] + dp + (bj[j] + db >= B ? 1u : 0u) : pj[j]];
                const uint64_t v16 = (uint64_t)bk * 64u + (uint32_t)lane;        // this lane's 16 bytes of the block (CPL cells x NV votes)
                const int4 q = stream_load(reinterpret_cast<const int4*>(a.answers) + v16);
                const uint32_t qq[4] = {(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w};
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int i = 0; i < NV; ++i) o.w[j][i] = qq[j * NV + i];
                if (TOK) {
                    const int4 y = stream_load(reinterpret_cast<const int4*>(a.tokens) + v16);
                    const int32_t yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                    for (int j = 0; j < CPL; ++j)
#pragma unroll
                        for (int i = 0; i < NV; ++i) o.tk[j][i] = yy[j * NV + i];
                }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const uint64_t c = (uint64_t)bk * BLK + 64u * j + (uint32_t)lane;
            if (pre) o.truth[j] = a.truth[ahead ? pj[j] + dp + (bj[j] + db >= B ? 1u : 0u) : pj[j]];
            if constexpr (NV == 1) {
                o.w[j][0] = (uint32_t)__builtin_nontemporal_load(a.answers + c);
                if (TOK) o.tk[j][0] = __builtin_nontemporal_load(a.tokens + c);
            } else if constexpr (NV == 2) {
                typedef int v2i32 __attribute__((ext_vector_type(2)));
                const v2i32 q = __builtin_nontemporal_load(reinterpret_cast<const v2i32*>(a.answers) + c);
                o.w[j][0] = (uint32_t)q.x; o.w[j][1] = (uint32_t)q.y;
                if (TOK) { const v2i32 y = __builtin_nontemporal_load(reinterpret_cast<const v2i32*>(a.tokens) + c); o.tk[j][0] = y.x; o.tk[j][1] = y.y; }
            } else {
                const int4 q = stream_load(reinterpret_cast<const int4*>(a.answers) + c);
                o.w[j][0] = (uint32_t)q.x; o.w[j][1] = (uint32_t)q.y; o.w[j][2] = (uint32_t)q.z; o.w[j][3] = (uint32_t)q.w;
                if (TOK) { const int4 y = stream_load(reinterpret_cast<const int4*>(a.tokens) + c); o.tk[j][0] = y.x; o.tk[j][1] = y.y; o.tk[j][2] = y.z; o.tk[j][3] = y.w; }
            }
        }
    };
    Votes cur{}, nxt{};
    if (blk < nblocks) load(blk, cur, false);
    for (; blk < nblocks; blk += nwaves) {
        if (blk + nwaves < nblocks) load(blk + nwaves, nxt, true);   // one step ahead
        uint32_t pk[CPL];                                            // (consec + packed records: the lane's CPL records leave in one store)
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const uint32_t b = bj[j];
            const uint32_t n = fixed_b ? nj[j] : (uint32_t)(valid_len(a, (int32_t)b) < NV ? valid_len(a, (int32_t)b) : NV);
            const int32_t truth = pre ? cur.truth[j] : a.truth[pj[j]];
            const uint32_t tcmp = (truth >= 0 && truth < kBins) ? (uint32_t)truth : 0xffffffffu;
            // every vote is a bin: an out-of-domain vote (o1.py:140 int(extracted_answer) is unbounded; the extractor maps it into bins 0..1023)
            // counts for bin 1023 -- and raises the error word when it is inside the valid prefix
            uint32_t w[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                bad |= (uint32_t)i < n ? cur.w[j][i] : 0u;
                w[i] = cur.w[j][i] < 1023u ? cur.w[j][i] : 1023u;
            }
            // statistics.multimode on <= 4 votes: count_i = #{ k valid : x_k == x_i }; a mode is counted at its FIRST occurrence
            uint32_t cnt[NV], maxc = 0, tc = 0;
            long long tok = 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool vi = (uint32_t)i < n;
                uint32_t c = 0;
#pragma unroll
                for (int k = 0; k < NV; ++k) c += ((uint32_t)k < n && w[k] == w[i]) ? 1u : 0u;
                cnt[i] = vi ? c : 0u;
                maxc = cnt[i] > maxc ? cnt[i] : maxc;
                tc += (vi && w[i] == tcmp) ? 1u : 0u;
                if (TOK) tok += vi ? (long long)cur.tk[j][i] : 0ll;
            }
            uint32_t n_modes = 0, mm = 0xffffu;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                bool first = true;
#pragma unroll
                for (int k = 0; k < i; ++k) first = first && !((uint32_t)k < n && w[k] == w[i]);
                const bool mode = cnt[i] == maxc && maxc > 0 && first;
                n_modes += mode ? 1u : 0u;
                mm = (mode && w[i] < mm) ? w[i] : mm;
            }
            const bool any = maxc > 0;
            const uint32_t hit = (any && tc == maxc) ? 1u : 0u;                   // o1.py:206
            const uint64_t c = slot_cell(blk, j);
            pk[j] = pack_cell(maxc, tc, n_modes, mm, hit);
            if (a.cells) {
                if (a.packed_cells) { if (NV == 4) __builtin_nontemporal_store(pk[j], reinterpret_cast<uint32_t*>(a.cells) + c); }
                else __builtin_nontemporal_store(scv_v4u{maxc, tc, (n_modes & 0xffffu) | ((any ? mm : 0xffffu) << 16), hit}, reinterpret_cast<scv_v4u*>(a.cells) + c);
            }
            if (TOK && a.cell_tokens) a.cell_tokens[c] = tok;
            if (fixed_b) {                                                        // o1.py:238-240 as integers, in registers
                h1[j] += (hit && n_modes == 1u) ? 1u : 0u;
                if (counters && hit && n_modes != 1u) atomicAdd(&tie[b * TC + n_modes], 1u);
                tcs[j] += tc;
                if (TOK) toks[j] += tok;
            } else if (counters) {
                if (hit) atomicAdd(&tie[b * TC + n_modes], 1u);
                if (tc) atomicAdd(&acc[b], (unsigned long long)tc);
                if (TOK) atomicAdd(&acc[B + b], (unsigned long long)tok);
            }
            pj[j] += dp; bj[j] += db;
            if (bj[j] >= B) { bj[j] -= B; pj[j] += 1u; }
        }
        if constexpr (NV < 4) {
            if (a.cells && a.packed_cells) {                         // (consec: 4 / 2 consecutive 4-byte records per lane)
                uint32_t* const out = reinterpret_cast<uint32_t*>(a.cells) + (uint64_t)blk * BLK + (uint32_t)CPL * (uint32_t)lane;
                if constexpr (NV == 1) __builtin_nontemporal_store(scv_v4u{pk[0], pk[1], pk[2], pk[3]}, reinterpret_cast<scv_v4u*>(out));
                else __builtin_nontemporal_store(scv_v2u{pk[0], pk[1]}, reinterpret_cast<scv_v2u*>(out));
            }
        }
        cur = nxt;
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    if (counters && fixed_b) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            if (h1[j]) atomicAdd(&tie[bj[j] * TC + 1], h1[j]);
            if (tcs[j]) atomicAdd(&acc[bj[j]], tcs[j]);
            if (TOK && toks[j]) atomicAdd(&acc[B + bj[j]], (unsigned long long)toks[j]);
        }
    }
    if (counters) {
        __syncthreads();
        for (int64_t i = tid; i < (int64_t)a.B * TC; i += T) {
            const uint32_t c = tie[i];
            if (c && a.tie_hits) {
                const int64_t b = i / TC;
                atomicAdd(&a.tie_hits[b * SCV_TIE_CLASSES + (i - b * TC)], (unsigned long long)c);
            }
        }
        for (int i = tid; i < a.B; i += T) {
            if (a.truth_sum && acc[i]) atomicAdd(&a.truth_sum[i], acc[i]);
            if (TOK && a.token_sum && acc[B + i]) atomicAdd(&a.token_sum[i], acc[B + i]);
        }
    }
}


// ---- kernel 1f1: cells of exactly ONE vote -- the reference's most common call (round 6) -------------------------------------------
//
// o1.py:302 runs every ask-nicely budget with N = 1, o1.py:276 the first eight majority budgets: a cell IS its vote.  statistics.multimode([x]) = [x]:
// max_count = 1, one mode, min_mode = x, truth_count = hit = (x == truth); an empty cell (n_valid = 0) has no mode.  scv_few_votes<1> serves such cells
// with its general machinery -- 62 VALU + 47 SALU per cell in the ISA (tools: the main loop of scv_few_votes<1, false>): the launch is bound by its
// instructions, not by HBM (counters only: 139 us for 1.02e8 cells = 2.9 TB/s).  Here the loop body of a cell is a clamp, one compare and one
// add: hits with one mode and truth votes are the SAME number at N = 1, so one 32-bit accumulator per cell slot feeds both tie_hits[b][1] and
// truth_sum[b].  A wave takes 256 consecutive cells per step; with 16-byte records lane l owns block + l, block + 64 + l ... (every store instruction
// of the wave touches consecutive bytes), without them -- counters only, or the 4-byte records of SCV_FLAG_PACKED_CELLS -- lane l owns the four
// consecutive cells block + 4 l ...: ONE 16-byte load, one 16-byte store of four packed records.  The last ncells % 256 cells (no whole block) are
// taken one per lane by one wave, with guarded loads.  Host contract: N == 1, 16-byte aligned bases, ncells < 2^29, no pool rows, and
// (grid * 16 * 256) % B == 0: a lane's cell slots keep their budgets for the whole launch.
template <bool TOK>
__global__ __launch_bounds__(1024) void scv_one_vote(const AggArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds);        // [B] hits = truth votes | [B] token sums
    const int tid = threadIdx.x, T = (int)blockDim.x, lane = tid & 63;
    const uint32_t B = (uint32_t)a.B;
    const bool counters = a.tie_hits || a.truth_sum || (TOK && a.token_sum);
    if (counters) {
        for (int i = tid; i < 2 * (int)B; i += T) acc[i] = 0;
        __syncthreads();
    }
    const bool consec = !a.cells || a.packed_cells;
    const uint32_t nblocks = (uint32_t)(a.ncells >> 8);
    const uint32_t nwaves = (uint32_t)gridDim.x * (uint32_t)(T >> 6);
    uint32_t blk = (uint32_t)blockIdx.x * (uint32_t)(T >> 6) + (uint32_t)(tid >> 6);
    const uint32_t dp = (nwaves * 256u) / B;                         // (the host rounds the grid: nwaves * 256 is a multiple of B)
    uint32_t pj[4], nj[4], bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = blk * 256u + (consec ? 4u * (uint32_t)lane + (uint32_t)j : 64u * (uint32_t)j + (uint32_t)lane);
        pj[j] = c / B;
        bj[j] = c - pj[j] * B;
        nj[j] = valid_len(a, (int32_t)bj[j]) > 0 ? 0xffffffffu : 0u;              // all ones: the cell has its vote
    }
    uint32_t hits[4] = {0u, 0u, 0u, 0u};
    long long toks[4] = {0ll, 0ll, 0ll, 0ll};
    uint32_t bad = 0;
    struct Step { uint32_t w[4]; int32_t tk[4]; int32_t truth[4]; };
    auto load = [&](uint32_t bk, Step& o, uint32_t pstep) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o.truth[j] = a.truth[pj[j] + pstep];
        if (consec) {
            const int4 q = stream_load(reinterpret_cast<const int4*>(a.answers) + ((uint64_t)bk * 64u + (uint32_t)lane));
            o.w[0] = (uint32_t)q.x; o.w[1] = (uint32_t)q.y; o.w[2] = (uint32_t)q.z; o.w[3] = (uint32_t)q.w;
            if (TOK) {
                const int4 y = stream_load(reinterpret_cast<const int4*>(a.tokens) + ((uint64_t)bk * 64u + (uint32_t)lane));
                o.tk[0] = y.x; o.tk[1] = y.y; o.tk[2] = y.z; o.tk[3] = y.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t c = (uint64_t)bk * 256u + 64u * (uint32_t)j + (uint32_t)lane;
                o.w[j] = (uint32_t)__builtin_nontemporal_load(a.answers + c);
                if (TOK) o.tk[j] = __builtin_nontemporal_load(a.tokens + c);
            }
        }
    };
    // the cells behind the last whole block: one per lane, by one wave, BEFORE its main loop (here few registers are live: behind the loop this
    // code cost the votes-only kernel 4 VGPRs = its second workgroup per CU)
    const uint32_t tail = (uint32_t)(a.ncells & 255);
    if (tail && blk == nblocks % nwaves) {
        for (uint32_t i = (uint32_t)lane; i < tail; i += 64u) {
            const uint32_t c = (nblocks << 8) + i;                  // (ncells < 2^29: 32-bit arithmetic)
            const uint32_t p = c / B, b = c - p * B;
            const uint32_t have = valid_len(a, (int32_t)b) > 0 ? 0xffffffffu : 0u;
            const uint32_t x = (uint32_t)a.answers[c];
            bad |= x & have;
            const uint32_t w = x < 1023u ? x : 1023u;
            const uint32_t tc = ((uint32_t)a.truth[p] == w ? 1u : 0u) & have;
            long long tv = 0;
            if (TOK) tv = have ? (long long)a.tokens[c] : 0ll;
            if (a.cells) {
                if (a.packed_cells) reinterpret_cast<uint32_t*>(a.cells)[c] = have ? (1u | (tc << 7) | (1u << 14) | (w << 21) | (tc << 31)) : (0x3ffu << 21);
                else reinterpret_cast<scv_v4u*>(a.cells)[c] = scv_v4u{have & 1u, tc, have ? (1u | (w << 16)) : 0xffff0000u, tc};
            }
            if (TOK && a.cell_tokens) a.cell_tokens[c] = tv;
            if (counters) {
                if (tc) atomicAdd(&acc[b], 1ull);
                if (TOK && tv) atomicAdd(&acc[B + b], (unsigned long long)tv);
            }
        }
    }
    Step cur{}, nxt{};
    if (blk < nblocks) load(blk, cur, 0u);
    for (; blk < nblocks; blk += nwaves) {
        if (blk + nwaves < nblocks) load(blk + nwaves, nxt, dp);     // one step ahead (its problems: this step's + dp)
        uint32_t pk[4];
        long long tv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t have = nj[j];
            bad |= cur.w[j] & have;
            const uint32_t w = cur.w[j] < 1023u ? cur.w[j] : 1023u;                // (an out-of-domain vote counts for bin 1023 and raises the error word)
            const uint32_t tc = ((uint32_t)cur.truth[j] == w ? 1u : 0u) & have;    // o1.py:206 (a truth outside the bins equals no clamped vote ... except
            hits[j] += tc;                                                         //  1023 itself, which IS bin 1023: the same rule as every kernel)
            if (TOK) { tv[j] = have ? (long long)cur.tk[j] : 0ll; toks[j] += tv[j]; }
            if (a.cells) {
                if (a.packed_cells) pk[j] = have ? (1u | (tc << 7) | (1u << 14) | (w << 21) | (tc << 31)) : (0x3ffu << 21);
                else {
                    const uint64_t c = (uint64_t)blk * 256u + 64u * (uint32_t)j + (uint32_t)lane;
                    __builtin_nontemporal_store(scv_v4u{have & 1u, tc, have ? (1u | (w << 16)) : 0xffff0000u, tc}, reinterpret_cast<scv_v4u*>(a.cells) + c);
                }
            }
            pj[j] += dp;
        }
        if (a.cells && a.packed_cells)
            __builtin_nontemporal_store(scv_v4u{pk[0], pk[1], pk[2], pk[3]}, reinterpret_cast<scv_v4u*>(reinterpret_cast<uint32_t*>(a.cells) + (uint64_t)blk * 256u) + (uint32_t)lane);
        if (TOK && a.cell_tokens) {
            if (consec) {
                scv_v4u* const out = reinterpret_cast<scv_v4u*>(a.cell_tokens + (uint64_t)blk * 256u + 4u * (uint32_t)lane);
                out[0] = scv_v4u{(uint32_t)tv[0], (uint32_t)((unsigned long long)tv[0] >> 32), (uint32_t)tv[1], (uint32_t)((unsigned long long)tv[1] >> 32)};
                out[1] = scv_v4u{(uint32_t)tv[2], (uint32_t)((unsigned long long)tv[2] >> 32), (uint32_t)tv[3], (uint32_t)((unsigned long long)tv[3] >> 32)};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a.cell_tokens[(uint64_t)blk * 256u + 64u * (uint32_t)j + (uint32_t)lane] = tv[j];
            }
        }
        cur = nxt;
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    if (counters) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (hits[j]) atomicAdd(&acc[bj[j]], (unsigned long long)hits[j]);
            if (TOK && toks[j]) atomicAdd(&acc[B + bj[j]], (unsigned long long)toks[j]);
        }
        __syncthreads();
        for (int i = tid; i < (int)B; i += T) {
            if (acc[i]) {
                if (a.tie_hits) atomicAdd(&a.tie_hits[(int64_t)i * SCV_TIE_CLASSES + 1], acc[i]);
                if (a.truth_sum) atomicAdd(&a.truth_sum[i], acc[i]);
            }
            if (TOK && a.token_sum && acc[B + i]) atomicAdd(&a.token_sum[i], acc[B + i]);
        }
    }
}

// ---- kernel 1f2: cells of exactly TWO votes (o1.py:276: the budget T = 4096) -- the same idea as scv_one_vote ---------------------------------
//
// multimode([x, y]) is [x] when x == y and [x, y] otherwise: max_count = 1 + (x == y), modes = 2 - (x == y), min_mode = min(x, y),
// truth_count = (x == t) + (y == t), hit = (truth_count == max_count) -- a budget that sees only the first vote (n_valid = 1) is the one-vote case on x.
// Two 32-bit accumulators per cell slot (hits with one mode, hits with two) + the truth votes.  A wave takes 128 consecutive cells per step: with
// 16-byte records lane l owns block + l and block + 64 + l (two 8-byte loads), without them the two consecutive cells block + 2 l, + 1 (one
// 16-byte load, one 8-byte store of two packed records).  Host contract as scv_one_vote's with blocks of 128 cells.
template <bool TOK>
__global__ __launch_bounds__(1024) void scv_two_votes(const AggArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds);        // [B] hits, one mode | [B] hits, two modes | [B] truth votes | [B] token sums
    const int tid = threadIdx.x, T = (int)blockDim.x, lane = tid & 63;
    const uint32_t B = (uint32_t)a.B;
    const bool counters = a.tie_hits || a.truth_sum || (TOK && a.token_sum);
    if (counters) {
        for (int i = tid; i < 4 * (int)B; i += T) acc[i] = 0;
        __syncthreads();
    }
    const bool consec = !a.cells || a.packed_cells;
    const uint32_t nblocks = (uint32_t)(a.ncells >> 7);
    const uint32_t nwaves = (uint32_t)gridDim.x * (uint32_t)(T >> 6);
    uint32_t blk = (uint32_t)blockIdx.x * (uint32_t)(T >> 6) + (uint32_t)(tid >> 6);
    const uint32_t dp = (nwaves * 128u) / B;                         // (the host rounds the grid: nwaves * 128 is a multiple of B)
    uint32_t pj[2], bj[2], m0[2], m1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t c = blk * 128u + (consec ? 2u * (uint32_t)lane + (uint32_t)j : 64u * (uint32_t)j + (uint32_t)lane);
        pj[j] = c / B;
        bj[j] = c - pj[j] * B;
        const int64_t n = valid_len(a, (int32_t)bj[j]);
        m0[j] = n >= 1 ? 0xffffffffu : 0u;
        m1[j] = n >= 2 ? 0xffffffffu : 0u;
    }
    uint32_t h1[2] = {0u, 0u}, h2[2] = {0u, 0u}, tcs[2] = {0u, 0u};
    long long toks[2] = {0ll, 0ll};
    uint32_t bad = 0;
    // one cell: everything the record and the counters need
    struct Out { uint32_t maxc, tc, n_modes, mm, hit; };
    auto count = [&](uint32_t x0, uint32_t x1, uint32_t t, uint32_t k0, uint32_t k1) -> Out {
        bad |= (x0 & k0) | (x1 & k1);
        const uint32_t w0 = x0 < 1023u ? x0 : 1023u, w1 = x1 < 1023u ? x1 : 1023u;
        const uint32_t eq = (w0 == w1 ? 1u : 0u) & k1;
        Out o;
        o.maxc = (1u + eq) & k0;
        o.n_modes = ((k1 & 1u) + 1u - eq) & k0;
        o.mm = (k1 && w1 < w0) ? w1 : w0;
        o.tc = ((t == w0 ? 1u : 0u) & k0) + ((t == w1 ? 1u : 0u) & k1);
        o.hit = (o.tc == o.maxc ? 1u : 0u) & k0;
        return o;
    };
    // the cells behind the last whole block: one per lane, by one wave, before its main loop (few registers are live here)
    const uint32_t tail = (uint32_t)(a.ncells & 127);
    if (tail && blk == nblocks % nwaves) {
        for (uint32_t i = (uint32_t)lane; i < tail; i += 64u) {
            const uint32_t c = (nblocks << 7) + i;
            const uint32_t p = c / B, b = c - p * B;
            const int64_t n = valid_len(a, (int32_t)b);
            const uint32_t k0 = n >= 1 ? 0xffffffffu : 0u, k1 = n >= 2 ? 0xffffffffu : 0u;
            const Out o = count((uint32_t)a.answers[2u * (uint64_t)c], (uint32_t)a.answers[2u * (uint64_t)c + 1u], (uint32_t)a.truth[p], k0, k1);
            long long tv = 0;
            if (TOK) tv = (k0 ? (long long)a.tokens[2u * (uint64_t)c] : 0ll) + (k1 ? (long long)a.tokens[2u * (uint64_t)c + 1u] : 0ll);
            if (a.cells) {
                if (a.packed_cells) reinterpret_cast<uint32_t*>(a.cells)[c] = pack_cell(o.maxc, o.tc, o.n_modes, o.mm, o.hit);
                else reinterpret_cast<scv_v4u*>(a.cells)[c] = scv_v4u{o.maxc, o.tc, o.n_modes | ((k0 ? o.mm : 0xffffu) << 16), o.hit};
            }
            if (TOK && a.cell_tokens) a.cell_tokens[c] = tv;
            if (counters) {
                if (o.hit) atomicAdd(&acc[(o.n_modes == 2u ? B : 0u) + b], 1ull);
                if (o.tc) atomicAdd(&acc[2 * B + b], (unsigned long long)o.tc);
                if (TOK && tv) atomicAdd(&acc[3 * B + b], (unsigned long long)tv);
            }
        }
    }
    struct Step { uint32_t w[2][2]; int32_t tk[2][2]; int32_t truth[2]; };
    typedef int v2i32 __attribute__((ext_vector_type(2)));
    auto load = [&](uint32_t bk, Step& o, uint32_t pstep) {
#pragma unroll
        for (int j = 0; j < 2; ++j) o.truth[j] = a.truth[pj[j] + pstep];
        if (consec) {
            const int4 q = stream_load(reinterpret_cast<const int4*>(a.answers) + ((uint64_t)bk * 64u + (uint32_t)lane));
            o.w[0][0] = (uint32_t)q.x; o.w[0][1] = (uint32_t)q.y; o.w[1][0] = (uint32_t)q.z; o.w[1][1] = (uint32_t)q.w;
            if (TOK) {
                const int4 y = stream_load(reinterpret_cast<const int4*>(a.tokens) + ((uint64_t)bk * 64u + (uint32_t)lane));
                o.tk[0][0] = y.x; o.tk[0][1] = y.y; o.tk[1][0] = y.z; o.tk[1][1] = y.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint64_t c = (uint64_t)bk * 128u + 64u * (uint32_t)j + (uint32_t)lane;
                const v2i32 q = __builtin_nontemporal_load(reinterpret_cast<const v2i32*>(a.answers) + c);
                o.w[j][0] = (uint32_t)q.x; o.w[j][1] = (uint32_t)q.y;
                if (TOK) { const v2i32 y = __builtin_nontemporal_load(reinterpret_cast<const v2i32*>(a.tokens) + c); o.tk[j][0] = y.x; o.tk[j][1] = y.y; }
            }
        }
    };
    Step cur{}, nxt{};
    if (blk < nblocks) load(blk, cur, 0u);
    for (; blk < nblocks; blk += nwaves) {
        if (blk + nwaves < nblocks) load(blk + nwaves, nxt, dp);     // one step ahead
        uint32_t pk[2];
        long long tv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const Out o = count(cur.w[j][0], cur.w[j][1], (uint32_t)cur.truth[j], m0[j], m1[j]);
            h1[j] += o.hit & (o.n_modes == 1u ? 1u : 0u);
            h2[j] += o.hit & (o.n_modes == 2u ? 1u : 0u);
            tcs[j] += o.tc;
            if (TOK) { tv[j] = (m0[j] ? (long long)cur.tk[j][0] : 0ll) + (m1[j] ? (long long)cur.tk[j][1] : 0ll); toks[j] += tv[j]; }
            if (a.cells) {
                if (a.packed_cells) pk[j] = pack_cell(o.maxc, o.tc, o.n_modes, o.mm, o.hit);
                else {
                    const uint64_t c = (uint64_t)blk * 128u + 64u * (uint32_t)j + (uint32_t)lane;
                    __builtin_nontemporal_store(scv_v4u{o.maxc, o.tc, o.n_modes | ((m0[j] ? o.mm : 0xffffu) << 16), o.hit}, reinterpret_cast<scv_v4u*>(a.cells) + c);
                }
            }
            pj[j] += dp;
        }
        if (a.cells && a.packed_cells)
            __builtin_nontemporal_store(scv_v2u{pk[0], pk[1]}, reinterpret_cast<scv_v2u*>(reinterpret_cast<uint32_t*>(a.cells) + (uint64_t)blk * 128u) + (uint32_t)lane);
        if (TOK && a.cell_tokens) {
            if (consec) {
                scv_v4u* const out = reinterpret_cast<scv_v4u*>(a.cell_tokens + (uint64_t)blk * 128u + 2u * (uint32_t)lane);
                out[0] = scv_v4u{(uint32_t)tv[0], (uint32_t)((unsigned long long)tv[0] >> 32), (uint32_t)tv[1], (uint32_t)((unsigned long long)tv[1] >> 32)};
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) a.cell_tokens[(uint64_t)blk * 128u + 64u * (uint32_t)j + (uint32_t)lane] = tv[j];
            }
        }
        cur = nxt;
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    if (counters) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (h1[j]) atomicAdd(&acc[bj[j]], (unsigned long long)h1[j]);
            if (h2[j]) atomicAdd(&acc[B + bj[j]], (unsigned long long)h2[j]);
            if (tcs[j]) atomicAdd(&acc[2 * B + bj[j]], (unsigned long long)tcs[j]);
            if (TOK && toks[j]) atomicAdd(&acc[3 * B + bj[j]], (unsigned long long)toks[j]);
        }
        __syncthreads();
        for (int i = tid; i < (int)B; i += T) {
            if (a.tie_hits) {
                if (acc[i]) atomicAdd(&a.tie_hits[(int64_t)i * SCV_TIE_CLASSES + 1], acc[i]);
                if (acc[B + i]) atomicAdd(&a.tie_hits[(int64_t)i * SCV_TIE_CLASSES + 2], acc[B + i]);
            }
            if (a.truth_sum && acc[2 * B + i]) atomicAdd(&a.truth_sum[i], acc[2 * B + i]);
            if (TOK && a.token_sum && acc[3 * B + i]) atomicAdd(&a.token_sum[i], acc[3 * B + i]);
        }
    }
}

// Per-workgroup accumulation of the per-budget counters (o1.py:238-240 as integers) for the register-resident
// kernels: [B][TCL] tie-class hits (u32) and [B] truth-vote | [B] token sums (u64) live behind the waves' private
// regions; one lane per cell adds to them with LDS atomics and the workgroup flushes the non-zero words with one
// device atomic each when it has no cells left.  Replaces the separate scv_reduce_cells launch (12-15 us at
// 10^5-10^6 cells) and, when the caller wants no cell table, every cell write.
struct WgCounters {
    uint32_t* tie;
    unsigned long long* sums;
    int32_t tcl;
};
__device__ __forceinline__ WgCounters wg_counters_begin(const AggArgs& a, uint32_t* region, int tid, int nthreads) {
    WgCounters w;
    w.tcl = a.acc_classes;
    w.tie = region;
    const int64_t tie_words = ((int64_t)a.B * w.tcl + 1) & ~(int64_t)1;
    w.sums = reinterpret_cast<unsigned long long*>(region + tie_words);
    if (w.tcl > 0) {
        for (int64_t i = tid; i < tie_words + 4 * (int64_t)a.B; i += nthreads) region[i] = 0;
        __syncthreads();
    }
    return w;
}
template <bool TOK>
__device__ __forceinline__ void wg_counters_add(const AggArgs& a, const WgCounters& w, int32_t b, uint32_t hit, uint32_t n_modes,
                                                uint32_t tc, long long tok) {
    if (hit) {
        if ((int32_t)n_modes < w.tcl) atomicAdd(&w.tie[(int64_t)b * w.tcl + (int32_t)n_modes], 1u);
        else if (a.tie_hits) atomicAdd(&a.tie_hits[(int64_t)b * SCV_TIE_CLASSES + n_modes], 1ull);
    }
    if (tc) atomicAdd(&w.sums[b], (unsigned long long)tc);
    if (TOK) atomicAdd(&w.sums[a.B + b], (unsigned long long)tok);
}
template <bool TOK>
__device__ __forceinline__ void wg_counters_flush(const AggArgs& a, const WgCounters& w, int tid, int nthreads) {
    if (w.tcl <= 0) return;
    __syncthreads();
    if (a.tie_hits) {
        for (int64_t i = tid; i < (int64_t)a.B * w.tcl; i += nthreads) {
            const uint32_t v = w.tie[i];
            if (v) {
                const int64_t b = i / w.tcl;
                atomicAdd(&a.tie_hits[b * SCV_TIE_CLASSES + (i - b * w.tcl)], (unsigned long long)v);
            }
        }
    }
    for (int i = tid; i < a.B; i += nthreads) {
        if (a.truth_sum && w.sums[i]) atomicAdd(&a.truth_sum[i], w.sums[i]);
        if (TOK && a.token_sum && w.sums[a.B + i]) atomicAdd(&a.token_sum[i], w.sums[a.B + i]);
    }
}

// ---- kernel 1f'': prefix budgets over SHORT pools, one lane per PROBLEM, online mode tracking ------
//
// The reference's own shape (o1.py:274-277): maj@1, 2, 4 ... N over ONE pool of N <= 64 samples per problem.  A lane
// owns a problem and feeds its votes one at a time, in order, into a running (max_count, n_modes, min_mode, truth
// votes): vote i of value x has count c = 1 + #{ j < i : x_j == x } (i register compares), and adding it changes the
// mode statistics exactly one way --
//     c >  max_count : x is the new unique mode          c == max_count : x joins the modes          else nothing
// -- so the state after vote i IS statistics.multimode's answer for the prefix 0..i, and a budget is a snapshot of the
// state at its boundary: every budget of a problem comes out of ONE pass over its N votes (N (N - 1) / 2 compares: 2016
// at N = 64) instead of one count per budget.  Boundaries are visited in ascending order (rank sort of n_valid in
// LDS; unsorted / duplicate / empty budgets allowed); the per-budget counters accumulate in LDS as in scv_lane_cells.
// TB: launch bound (the block may be smaller).  A boundary only stores the lane's snapshot (16-byte cell record, token sum) in
// LDS, and the counters of all B budgets are taken from the snapshots after the last vote, then the wave's 64 x B records
// leave as one contiguous block -- the unrolled vote loop then carries 64 tiny snapshot sites instead of 64 copies of the
// reductions (with tokens the latter did not fully unroll and put the token registers in scratch: 230 us).  Budget lists
// whose snapshots do not fit the LDS (24 B x 64 x B per wave with tokens) run on scv_prefix_pool (round 5; rounds 2-4 kept
// a second form of this kernel with the reductions at every boundary for them).
template <int NV, int TB, bool TOK>
__global__ __launch_bounds__(TB) void scv_lane_prefix(const AggArgs a) {
    constexpr int TC = NV + 1;                                       // tie classes 0..NV
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int T = (int)blockDim.x;
    uint32_t* tie = lds;                                             // [B][TC]
    const int64_t tie_words = ((int64_t)a.B * TC + 1) & ~(int64_t)1;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds + tie_words);   // [B] truth sums | [B] token sums
    int32_t* ord = reinterpret_cast<int32_t*>(lds + tie_words + 4 * (int64_t)a.B);      // [B] budgets by ascending n_valid
    int32_t* nvs = ord + a.B;                                                            // [B] their n_valid, ascending
    // [waves][64 * B] snapshots of the wave's current 64 problems, in the order the cells have in memory
    uint4* stage_base = reinterpret_cast<uint4*>(lds + ((tie_words + 6 * (int64_t)a.B + 3) & ~(int64_t)3));
    uint4* stage_rec = stage_base + (int64_t)(threadIdx.x >> 6) * 64 * a.B;
    long long* stage_tok = reinterpret_cast<long long*>(stage_base + (int64_t)(T >> 6) * 64 * a.B) + (int64_t)(threadIdx.x >> 6) * 64 * a.B;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (sort_prefix_took_it(a, tid, T)) return;
    for (int64_t i = tid; i < tie_words + 4 * (int64_t)a.B; i += T) lds[i] = 0;
    for (int b0 = 0; b0 < a.B; b0 += T) {
        const int b = b0 + tid;
        const bool have = b < a.B;
        const int64_t nb = have ? valid_len(a, b) : 0;
        const int rank = budget_rank<false>(a, b, nb);
        if (have) { ord[rank] = b; nvs[rank] = (int32_t)nb; }
    }
    __syncthreads();
    const bool vec = a.wave_lds_words != 0;                          // host: N % 4 == 0 and 16-byte aligned bases
    const int32_t N = (int32_t)a.N;
    const int32_t B = a.B;
    uint32_t bad = 0;
    // the counters of budget b from every lane's (hit, n_modes, truth votes, tokens), reduced over the wave first: 64 lanes
    // adding to the same LDS word would be serialised 64 deep, three times per budget
    auto count_budget = [&](int32_t b, uint32_t hit, uint32_t n_modes, uint32_t tc, long long tok) {
        const unsigned long long m1 = __ballot(hit != 0u && n_modes == 1u);       // o1.py:238-240 as integers
        if (hit && n_modes != 1u) atomicAdd(&tie[b * TC + (int32_t)n_modes], 1u);
        const uint32_t tcs = wave_sum_u32(tc);
        long long toks = 0;
        if (TOK) toks = wave_sum_i64(tok);
        if (lane == 0) {
            if (m1) atomicAdd(&tie[b * TC + 1], (uint32_t)__popcll(m1));
            if (tcs) atomicAdd(&acc[b], (unsigned long long)tcs);
            if (TOK) atomicAdd(&acc[B + b], (unsigned long long)toks);
        }
    };
    const int64_t stride = (int64_t)gridDim.x * T;
    const int64_t first = (int64_t)blockIdx.x * T + tid - lane;     // lanes of a wave run the same number of steps
    for (int64_t p0 = first; p0 < a.P; p0 += stride) {
        const int64_t p = p0 + lane;
        const bool live = p < a.P;
        const int64_t off = (live ? p : 0) * a.N;
        const int32_t* row = a.answers + off;
        const int32_t* trow = TOK ? a.tokens + off : nullptr;
        // next boundary in a scalar register: the per-vote test is then one s_cmp (an LDS read per vote otherwise)
        auto boundary = [&](int kk) -> int32_t { return kk < B ? __builtin_amdgcn_readfirstlane(nvs[kk]) : -1; };
        if (TOK) {
            // Token sums of the budgets in a pass of their own, BEFORE the votes are loaded: 64 token registers next to
            // 64 votes + 32 packed pairs meant 226 VGPRs (2 waves per SIMD); the running sum is snapshot into the staged
            // slots at the same boundaries the vote pass will visit.
            int32_t tkv[NV];
            if (vec) {
#pragma unroll
                for (int kq = 0; kq < NV / 4; ++kq) {
                    int4 y = make_int4(0, 0, 0, 0);
                    if (4 * kq < N) y = stream_load(reinterpret_cast<const int4*>(trow) + kq);
                    tkv[4 * kq] = y.x; tkv[4 * kq + 1] = y.y; tkv[4 * kq + 2] = y.z; tkv[4 * kq + 3] = y.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    tkv[i] = 0;
                    if (i < N) tkv[i] = __builtin_nontemporal_load(trow + i);
                }
            }
            long long ts = 0;
            int kt = 0;
            int32_t nn = boundary(0);
            while (nn == 0) { stage_tok[lane * B + __builtin_amdgcn_readfirstlane(ord[kt])] = 0; nn = boundary(++kt); }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (i < N) {
                    ts += (long long)tkv[i];
                    while (nn == i + 1) { stage_tok[lane * B + __builtin_amdgcn_readfirstlane(ord[kt])] = ts; nn = boundary(++kt); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                       // the vote loads start after the token registers are dead
        }
        uint32_t x[NV];
        if (vec) {
#pragma unroll
            for (int k = 0; k < NV / 4; ++k) {
                int4 q = make_int4(0, 0, 0, 0);
                if (4 * k < N) q = stream_load(reinterpret_cast<const int4*>(row) + k);
                x[4 * k] = (uint32_t)q.x; x[4 * k + 1] = (uint32_t)q.y; x[4 * k + 2] = (uint32_t)q.z; x[4 * k + 3] = (uint32_t)q.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                x[i] = 0;
                if (i < N) x[i] = (uint32_t)__builtin_nontemporal_load(row + i);
            }
        }
        const int32_t truth = a.truth[live ? p : 0];
        const uint32_t tcmp = (truth >= 0 && truth < kBins) ? (uint32_t)truth : 0xffffffffu;
        uint32_t maxc = 0, n_modes = 0, min_mode = 0xffffu, tc = 0;
        int k = 0;                                                   // next boundary (wave-uniform)
        auto emit = [&](int32_t b) {
            const bool any = maxc > 0;
            const uint32_t hit = (live && any && tc == maxc) ? 1u : 0u;     // o1.py:206
            uint4 rec;
            rec.x = maxc;
            rec.y = tc;
            rec.z = (n_modes & 0xffffu) | ((any ? (min_mode & 0xffffu) : 0xffffu) << 16);
            rec.w = hit;
            stage_rec[lane * B + b] = rec;                           // (an inactive lane's slot is never copied out or counted;
                                                                     //  its token sum was staged by the token pass)
        };
        int32_t next_n = boundary(0);
        while (next_n == 0) { emit(__builtin_amdgcn_readfirstlane(ord[k])); next_n = boundary(++k); }
        // votes packed two per register (values <= 1023): one xor + one saturating packed subtract + one dot product
        // compare TWO earlier votes with vote i and add the matches -- no carry chain through VCC (v_cmp + v_addc
        // costs two wait states per compare on gfx950)
        uint32_t xp[NV / 2];
#pragma unroll
        for (int m = 0; m < NV / 2; ++m) {
            if (2 * m < N) bad |= x[2 * m];
            if (2 * m + 1 < N) bad |= x[2 * m + 1];
            const uint32_t lo = x[2 * m] < 1023u ? x[2 * m] : 1023u, hi = x[2 * m + 1] < 1023u ? x[2 * m + 1] : 1023u;
            xp[m] = lo | (hi << 16);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i < N) {                                             // uniform
                const uint32_t xi = (i & 1) ? (xp[i >> 1] >> 16) : (xp[i >> 1] & 0xffffu);
                const uint32_t xi2 = xi | (xi << 16);
                // pairs 0 .. (i + 1) / 2 - 1 hold votes j <= i (i odd, itself included) or j < i (i even: itself adds 1)
                uint32_t c = (i & 1) ? 0u : 1u;
#pragma unroll
                for (int m = 0; m < (i + 1) / 2; ++m) {
                    const scv_v2h e = __builtin_elementwise_sub_sat(scv_v2h{1, 1}, __builtin_bit_cast(scv_v2h, xp[m] ^ xi2));   // 1 where equal
                    c = __builtin_amdgcn_udot2(e, scv_v2h{1, 1}, c, false);
                }
                const bool gt = c > maxc, eq = c == maxc;
                n_modes = gt ? 1u : n_modes + (eq ? 1u : 0u);
                min_mode = gt ? xi : ((eq && xi < min_mode) ? xi : min_mode);
                maxc = gt ? c : maxc;
                tc += xi == tcmp ? 1u : 0u;
                while (next_n == i + 1) { emit(__builtin_amdgcn_readfirstlane(ord[k])); next_n = boundary(++k); }
            }
        }
        // LDS operations of a wave are in order: the snapshots are complete here
        __builtin_amdgcn_wave_barrier();
        for (int32_t b = 0; b < B; ++b) {
            const uint4 rec = stage_rec[lane * B + b];
            long long tv = 0;
            if (TOK) tv = stage_tok[lane * B + b];
            count_budget(b, live ? rec.w : 0u, rec.z & 0xffffu, live ? rec.y : 0u, live ? tv : 0ll);
        }
        // the wave's block: cells[p0 * B .. (p0 + 64) * B), contiguous
        const int64_t nrec = (a.P - p0 < 64 ? a.P - p0 : 64) * B;
        if (a.cells) {
            uint4* out = reinterpret_cast<uint4*>(a.cells) + p0 * B;
            for (int64_t r = lane; r < nrec; r += 64) out[r] = stage_rec[r];
        }
        if (TOK && a.cell_tokens) {
            long long* out = reinterpret_cast<long long*>(a.cell_tokens) + p0 * B;
            for (int64_t r = lane; r < nrec; r += 64) out[r] = stage_tok[r];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    __syncthreads();
    for (int64_t i = tid; i < (int64_t)a.B * TC; i += T) {
        const uint32_t v = tie[i];
        if (v && a.tie_hits) {
            const int64_t b = i / TC;
            atomicAdd(&a.tie_hits[b * SCV_TIE_CLASSES + (i - b * TC)], (unsigned long long)v);
        }
    }
    for (int i = tid; i < a.B; i += T) {
        if (a.truth_sum && acc[i]) atomicAdd(&a.truth_sum[i], acc[i]);
        if (TOK && a.token_sum && acc[a.B + i]) atomicAdd(&a.token_sum[i], acc[a.B + i]);
    }
}

// ---- kernel 1g: register-resident cells (32 < N <= 4096), no barrier, no fold ---------------------
//
// PMC of the round-1 kernels in this range (profiles/r02_regimes_pmc_baseline.md): the wave-per-cell
// kernels spend 37-76 % of their wave cycles parked on memory (every pass re-reads the cell, one cell
// in flight per wave) and ~100 VALU + 40 SALU instructions of fixed work per cell; the streaming kernel
// pays a 1024*R-word fold, three workgroup barriers and ~9000 SALU instructions per wave at N = 1-4 K.
//
// Here a cell lives in REGISTERS: it occupies G = 16 / 32 / 64 adjacent lanes (C = 64/G cells per wave)
// and every lane holds V 16-byte vectors of it (capacity 4*G*V votes); the vectors of the NEXT batch of
// cells are loaded into a second register set before the current batch is counted, so a wave always
// has a whole batch (4 KiB) in flight and never waits on memory with nothing to do.  The waves of a
// workgroup are independent: no barrier exists in the loop.  Every cell slot of the wave owns a private
// histogram in LDS, 1024 bins x R copies with R = G/16, 16-bit counters (8 KiB per wave whatever G: the
// layouts are described at the kernel; a vote goes to copy lane % R), and only the bins a cell votes for
// are ever touched (sparse read-back, sparse clear):
//   pass 1  h[bin][copy] += 1                                (ds_add_u32 on the word holding the 16-bit counter;
//                                                             inactive vote slots add to per-lane trash words)
//   pass 2  key = (sum over copies of h[bin]) << 18 | address of the bin, in place of the vote; running max
//           -> one group reduction gives max_count AND the smallest modal bin
//   pass 3  #keys >= max_count << 18, divided by max_count = len(statistics.multimode); truth votes
//   pass 4  h[bin][copy] = 0 for every vote (same loop as pass 3)
// LDS operations of one wave execute in order, so the passes need no waits between them.  The
// histogram is indexed by 1023 - bin (a smaller bin has the larger address), so the maximum key is the
// smallest modal bin.

template <int G>
__device__ __forceinline__ uint32_t cellgroup_max(uint32_t v) {
    uint32_t t;
    t = dpp_mov<kQuadXor1>(v); v = v > t ? v : t;
    t = dpp_mov<kQuadXor2>(v); v = v > t ? v : t;
    t = dpp_mov<kHalfMirror>(v); v = v > t ? v : t;
    if (G >= 16) { t = dpp_mov<kRowMirror>(v); v = v > t ? v : t; }
    if (G >= 32) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = r[0] > r[1] ? r[0] : r[1]; }
    if (G >= 64) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = r[0] > r[1] ? r[0] : r[1]; }
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t cellgroup_sum(uint32_t v) {
    v += dpp_mov<kQuadXor1>(v);
    v += dpp_mov<kQuadXor2>(v);
    v += dpp_mov<kHalfMirror>(v);
    if (G >= 16) v += dpp_mov<kRowMirror>(v);
    if (G >= 32) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = r[0] + r[1]; }
    if (G >= 64) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = r[0] + r[1]; }
    return v;
}
template <int G>
__device__ __forceinline__ long long cellgroup_sum_i64(long long v) {    // limbs as in wave_sum_i64
    const unsigned long long u = (unsigned long long)v;
    const unsigned long long s0 = cellgroup_sum<G>((uint32_t)(u & 0x3fffffu));
    const unsigned long long s1 = cellgroup_sum<G>((uint32_t)((u >> 22) & 0x1fffffu));
    const unsigned long long s2 = cellgroup_sum<G>((uint32_t)((u >> 43) & 0x1fffffu));
    return (long long)(s0 + (s1 << 22) + (s2 << 43));
}

// key = count << kKeyShift | LDS byte address: a workgroup may hold all 160 KiB of a CU's LDS (18 address bits); counts are <= 4096 (13 bits)
constexpr int kKeyShift = 18;
constexpr uint32_t kKeyMask = (1u << kKeyShift) - 1u;
constexpr int kRegLaneWords = 64 + 128 + 64;         // + per lane: the first pivot's word, 8 bytes of trash (inactive vote slots add there: a
                                                     // shared trash bin serialised them 32-64 deep), the second pivot's word
// The waves of a workgroup are independent; a workgroup is ALL the waves a CU holds of the shape (launch bounds =
// the occupancy the shape is meant to run at: 16 / 12 / 8 / 4 waves), so a launch is one workgroup per CU: the
// end-of-launch counter flush then costs 256 device atomics per counter (12 ns each on one address), not 1024.
template <int G, int V, bool TOK, bool VEC = true>
constexpr int reg_cells_waves() {
#ifdef SCV_G8_WAVES
    if (G == 8) return SCV_G8_WAVES;           // (A/B builds)
#endif
    // 8 lanes per cell, 8-bit bins: 8 cells x 1026 bytes = the same 8 KiB per wave.  (8, 3) runs K = 2 batches per iteration at 12 waves (136-157
    // VGPRs; measured on one box, N = 96, D1 / D0 / D3 / D5: 80.6 / 74.9 / 75.9 / 72.1 us against 82.7 / 78.8 / 79.5 / 77.3 with K = 1 at 16 waves)
    if (G == 8) return (V == 3 || TOK) ? 12 : 16;
    if (!VEC && V == 1) return 12;             // unaligned rows, K = 4 batches in flight: 12-28 B of scratch at 128 VGPRs
    // (with tokens the one-vector shapes run K = 4 batches per iteration, each with its token loads in flight: they
    // spill at 128 VGPRs -- tools/kernel_resources.py -- and get 12 waves = 168 VGPRs like the four-vector shapes)
    if (TOK && (V == 1 || V == 4)) return 12;
    if (G == 64 && V == 2 && TOK) return 12;
    return (V <= 2 || G < 64) ? 16 : 12;
}
template <int V, int H, bool TOK, bool VEC>
constexpr int reg_dense_waves() {
    if (V == 8) return TOK ? 4 : 8;   // (not instantiated any more)
    if (H == 1 || !VEC) return TOK ? 8 : 12;
    return 12;            // (126 VGPRs would allow 16 waves without tokens: measured slower, N = 4096 73 -> 82 us)
}
// 16-bit counters (sparse kernels; a cell slot holds <= 1024 votes): copy c of a cell is an array of 1026 u16 bins
// (1024 + trash + pad), so a wave's histograms take 8 KiB instead of 16 and twice the waves are resident
constexpr int kRegCopyBytes16 = 2 * 1026;
constexpr int kRegHist16Words = 4 * kRegCopyBytes16 / 4;
constexpr int kRegWaveWords16 = kRegHist16Words + kRegLaneWords;

// Per-vote state is ONE register holding the LDS byte address A of the vote's bin (all copies):
//   A = cellbase + ((1023 - bin) << S), S = log2(bytes between bins)  -> ds_add on the word of A (+ copy), reads at A,
//   key = count << 18 | A                                              -> clear at (key & 0x3ffff) (+ copy)
// so a full cell costs ~11.5 VALU per vote: or3 (domain, half) . min . mad . alignbyte . cmp . cndmask . and . ds_add |
// ds_read . lshl_or . max3 (half) | cmp . addc . and . ds_write.  A is < 2^18 (160 KiB of LDS), counts are <= 4096 < 2^13.
// LDS is addressed through address_space(3) pointers built from integers, so constant parts of an address
// land in the instruction's offset field instead of a VALU add.
//
// Votes for the cell's TRUTH value do not go to the histogram: each lane counts them in a word of its own (the
// "wide truth bin", 64 words per wave).  truth_count is needed anyway (pass@k's c, and the hit test), and the
// truth is the one value known per cell before looking at the data that is usually the hot one -- in the
// peaked distributions (and in the reference's real data whenever the majority is right) 40-70 % of a cell's
// votes are the truth, i.e. up to 16 lanes of one ds_add on the same copy of the same bin, serialised.
// Measured before this: peaked 2.6 / 3.3 / 4.3 / 5.5 TB/s at N = 256 / 1024 / 2048 / 4096 against 3.9 / 3.9 /
// 5.5 / 6.4 uniform.  The histogram then holds every value but the truth (whose bin stays 0) and the
// epilogue merges the two: max_count = max(hist max, truth_count), the truth joins the modes on a tie.
//
// DENSE (G = 64, long cells): after pass 1 the votes are dead; every lane scans its 16 bins (x = lane + 64 j:
// consecutive lanes read consecutive 16-byte slots, conflict-free ds_read_b128 with immediate offsets), so the
// per-vote cost is pass 1 alone (~6 instructions) and the rest is a fixed ~110 instructions per cell --
// cheaper than the sparse read-back from ~24 votes per lane up; len(multimode) is then a count of BINS.
// (scv_reg_cells keeps this as an A/B variant with 32-bit bins; the production dense scan is scv_reg_dense.)

__device__ __forceinline__ void lds_add(uint32_t addr, uint32_t inc) {
    __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>((uintptr_t)addr), inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
// 8-bit bins (G = 8): the increment of the byte A & 3 of its word: 1 << (8 * (A & 3)) = v_bfm_b32(1, A << 3) (bfm reads 5 bits of the shift)
__device__ __forceinline__ uint32_t byte_inc(uint32_t A) {
    uint32_t inc;
    asm("v_lshlrev_b32 %0, 3, %1\n\tv_bfm_b32 %0, 1, %0" : "=&v"(inc) : "v"(A));
    return inc;
}
__device__ __forceinline__ uint32_t sum_halves(uint32_t w, uint32_t acc) {      // acc + w.lo + w.hi: v_dot2_u32_u16
    return __builtin_amdgcn_udot2(__builtin_bit_cast(scv_v2h, w), scv_v2h{1, 1}, acc, false);
}
// 16-bit counters, the R copies of a bin packed side by side (R = 2: one word, R = 4: two words)
template <int R>
__device__ __forceinline__ uint32_t lds_count_packed(uint32_t A) {
    if (R == 4) { const scv_v2u q = *reinterpret_cast<lds_v2u*>((uintptr_t)A); return sum_halves(q.x, sum_halves(q.y, 0u)); }
    return sum_halves(*reinterpret_cast<lds_u32*>((uintptr_t)A), 0u);
}
// (Measured and dropped: one ds_wrxchg_rtn per vote that reads AND clears the bin -- no clearing pass, one key per
// distinct bin.  Uniform votes: N = 512 51 -> 45 us; but an exchange, unlike a read, serialises on equal addresses,
// and any popular value made it 2-4x slower: peaked 51 -> 112 us, exact ties 227 us.)
// sum over the R copies of a 16-bit bin: R ds_read_u16 at immediate offsets (copy stride kRegCopyBytes16)
template <int R>
__device__ __forceinline__ uint32_t lds_count16(uint32_t A) {
    uint32_t n = *reinterpret_cast<lds_u16*>((uintptr_t)A);
    if (R >= 2) n += *reinterpret_cast<lds_u16*>((uintptr_t)(A + kRegCopyBytes16));
    if (R == 4) { n += *reinterpret_cast<lds_u16*>((uintptr_t)(A + 2 * kRegCopyBytes16)); n += *reinterpret_cast<lds_u16*>((uintptr_t)(A + 3 * kRegCopyBytes16)); }
    return n;
}
// KB - (vm << S) in one VALU instruction (the compiler prefers shift + subtract)
template <int S>
__device__ __forceinline__ uint32_t bin_address(uint32_t KB, uint32_t vm) {
    uint32_t A;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(A) : "v"(vm), "n"(-(1 << S)), "v"(KB));
    return A;
}

// W = (A == ap0) ? tw : ((A == ap1) ? tw2 : A | c4): the pivot selection of one vote, hand-scheduled.  gfx950 wants two
// instructions between a VALU compare and the v_cndmask that reads its mask; left alone the compiler chains compare / s_nop /
// cndmask through VCC (seen in the ISA: 3 s_nop per vote, 4 idle cycles of ~36).  Here both compares are issued first -- one
// into an SGPR pair, one into VCC -- and an instruction the vote needs anyway (the address OR / the 16-bit increment) fills the gap.
__device__ __forceinline__ uint32_t pivot_select(uint32_t A, uint32_t c4, uint32_t ap0, uint32_t ap1, uint32_t tw, uint32_t tw2) {
    uint32_t W;
    unsigned long long m1;
    asm("v_cmp_eq_u32_e64 %[m1], %[a], %[p1]\n\t"
        "v_cmp_eq_u32_e32 vcc, %[a], %[p0]\n\t"
        "v_or_b32_e32 %[w], %[a], %[c]\n\t"
        "v_cndmask_b32_e64 %[w], %[w], %[t2], %[m1]\n\t"
        "v_cndmask_b32_e32 %[w], %[w], %[t1], vcc"
        : [w] "=&v"(W), [m1] "=&s"(m1)
        : [a] "v"(A), [c] "v"(c4), [p0] "v"(ap0), [p1] "v"(ap1), [t1] "v"(tw), [t2] "v"(tw2)
        : "vcc");
    return W;
}
// 16-bit bins, one copy (G = 16): the word that holds the bin (A & ~3) and the increment 1 or 1 << 16 by bit 1 of A
// (v_alignbyte_b32 shifts {1, 1} right by the low 2 bits of A in bytes: 0 or 2); a pivot's word collects the same increments.
__device__ __forceinline__ uint32_t pivot_select_h16(uint32_t A, uint32_t ap0, uint32_t ap1, uint32_t tw, uint32_t tw2, uint32_t& inc) {
    uint32_t W;
    unsigned long long m1;
    asm("v_cmp_eq_u32_e64 %[m1], %[a], %[p1]\n\t"
        "v_cmp_eq_u32_e32 vcc, %[a], %[p0]\n\t"
        "v_alignbyte_b32 %[i], 1, 1, %[a]\n\t"
        "v_cndmask_b32_e64 %[w], %[a], %[t2], %[m1]\n\t"
        "v_cndmask_b32_e32 %[w], %[w], %[t1], vcc\n\t"
        "v_and_b32_e32 %[w], -4, %[w]"
        : [w] "=&v"(W), [i] "=&v"(inc), [m1] "=&s"(m1)
        : [a] "v"(A), [p0] "v"(ap0), [p1] "v"(ap1), [t1] "v"(tw), [t2] "v"(tw2)
        : "vcc");
    return W;
}

// 8-bit bins (G = 8): the word that holds the bin (A & ~3) and the increment 1 << 8 (A & 3); a pivot's word collects the same increments
__device__ __forceinline__ uint32_t pivot_select_b8(uint32_t A, uint32_t ap0, uint32_t ap1, uint32_t tw, uint32_t tw2, uint32_t& inc) {
    uint32_t W;
    unsigned long long m1;
    asm("v_cmp_eq_u32_e64 %[m1], %[a], %[p1]\n\t"
        "v_cmp_eq_u32_e32 vcc, %[a], %[p0]\n\t"
        "v_lshlrev_b32 %[i], 3, %[a]\n\t"
        "v_cndmask_b32_e64 %[w], %[a], %[t2], %[m1]\n\t"
        "v_cndmask_b32_e32 %[w], %[w], %[t1], vcc\n\t"
        "v_bfm_b32 %[i], 1, %[i]\n\t"
        "v_and_b32_e32 %[w], -4, %[w]"
        : [w] "=&v"(W), [i] "=&v"(inc), [m1] "=&s"(m1)
        : [a] "v"(A), [p0] "v"(ap0), [p1] "v"(ap1), [t1] "v"(tw), [t2] "v"(tw2)
        : "vcc");
    return W;
}

// G lanes per cell, V 16-byte vectors per lane, K batches of C cells per loop iteration (short cells: more
// bytes in flight per wave), TOK tokens stream.  VEC: every row is 16-byte aligned (N % 4 == 0 and aligned bases).  !VEC: rows
// start anywhere (the reference's N is arbitrary, o1.py:276): a cell reads the 16-byte-aligned SUPERSET of its row with the same
// dwordx4 loads -- vector 0 starts `sh` = 0..3 elements before the row -- and masks both ends (votes are order-independent, so
// the shift only moves the validity window: slot e of the superset is vote e - sh).  The superset of n votes has up to n + 3
// slots; the host sizes the shape for N + 3.  (Round 2 used dword loads here: 3.45 vs 4.72 TB/s at N = 1001 / 1024.)
template <int G, int V, int K, bool TOK, bool VEC>
__global__ __launch_bounds__((64 * reg_cells_waves<G, V, TOK, VEC>())) void scv_reg_cells(const AggArgs a) {
    constexpr int C = 64 / G;                 // cells per wave per batch
    constexpr int R = G >= 16 ? G / 16 : 1;   // histogram copies per cell
    // 16-bit counters for every sparse shape (a cell slot holds <= 1024 votes): 8 KiB of LDS per wave instead of
    // 16, so 3-4 waves per SIMD are resident instead of 2 (measured +15-19 % at N = 64 ... 256).
    //  H16 (G = 16, one copy): a cell = 1026 u16 bins; a vote adds 1 << 16 (bin parity) to the WORD holding its bin
    //      (three more VALU per vote than 32-bit counters), reads and clears are 16-bit.
    //  P16 (G = 32 / 64, 2 / 4 copies): the copies of a bin sit side by side (one / two words per bin); a lane's
    //      increment (1 or 1 << 16) and word are constants of the lane, the read is one b32 / b64 + v_dot2_u32_u16.
    //      Two copies share a word, so all-equal votes serialise 32 deep instead of 16 (truth votes never do).
    //  B8  (G = 8, round 6: 65 ... 96 votes in a 96-slot shape -- 8 lanes x 3 vectors -- instead of the 128 slots of 16 x 2): a cell =
    //      1026 u8 bins, EIGHT cells per wave in the same 8 KiB; a vote adds 1 << 8 (A & 3) to the word holding its bin (one VALU more
    //      than H16: v_lshlrev + v_bfm instead of v_alignbyte), reads and clears are 8-bit.  A bin holds <= 255: cells of up to 128 votes.
    constexpr bool B8 = G == 8;
    constexpr bool H16 = G <= 16;             // one copy per cell, sub-word bins (B8 is the 8-bit form of it)
    constexpr bool P16 = G > 16;
    constexpr int S = B8 ? 0 : (H16 ? 1 : (R == 4 ? 3 : 2));   // log2(bytes between consecutive bins)
    constexpr int WW = kRegWaveWords16;
    constexpr uint32_t CELLBYTES = B8 ? 1026u : (uint32_t)(kRegCopyBytes16 * R);
    static_assert(!B8 || 4 * G * V <= 255, "8-bit bins hold at most 255 votes");
    constexpr int E = 4 * V;                  // votes per lane per batch
    constexpr uint32_t CAP = 4u * G * V;      // votes per cell slot
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_wg[];
    // The waves of a workgroup are independent (no barrier anywhere); they are grouped only so that the hardware
    // spreads them evenly over the 4 SIMDs: single-wave workgroups were measured to pile up (3+3+1+1), leaving
    // the kernel bound by its most crowded SIMD at ~75 % average residency.
    const int lane = threadIdx.x & 63;
    uint32_t* smem = smem_wg + (threadIdx.x >> 6) * a.wave_lds_words;
    const uint32_t base = (uint32_t)(uintptr_t)(lds_u32*)smem;   // LDS byte offset of this wave's region
    const int sub = lane / G, l = lane % G;
    {
        uint4* h4 = reinterpret_cast<uint4*>(smem);
        for (int i = lane; i < WW / 4; i += 64) h4[i] = make_uint4(0, 0, 0, 0);
    }
    const WgCounters wgc = wg_counters_begin(a, smem_wg + (blockDim.x >> 6) * a.wave_lds_words, (int)threadIdx.x, (int)blockDim.x);
    const uint32_t cellbase = base + (uint32_t)sub * CELLBYTES;   // bytes
    const uint32_t KB = cellbase + (1023u << S);        // address of bin 0 (histogram index 1023), copy 0
    const uint32_t LW = base + (uint32_t)(WW - kRegLaneWords) * 4u;          // first byte behind this wave's histograms
    const uint32_t ATR = LW + 256u + (uint32_t)lane * 8u;                    // this lane's trash (above every bin address)
    const uint32_t copy = (uint32_t)l & (R - 1);
    const uint32_t copy4 = P16 ? (copy >> 1) * 4u : copy * (uint32_t)kRegCopyBytes16;   // word of this lane's copy, bytes
    const uint32_t copy_inc = 1u << (16u * (copy & 1u));                 // P16: this lane's half of that word
    const uint32_t copy2 = copy * 2u;                                    // P16: byte offset of this lane's 16-bit counter
    const uint32_t TW = LW + (uint32_t)lane * 4u;                            // this lane's pivot words (first / second pivot)
    const uint32_t TW2 = LW + 768u + (uint32_t)lane * 4u;
    // n_valid[B] cached behind the histograms (host sizes the region; B > kMaxSortedB reads it from memory):
    // a global load here would put a dependent memory round trip in front of every batch's loads
    uint32_t* nv_lds = smem + WW;
    const bool nv_cached = a.n_valid && a.B <= kMaxSortedB;
    if (nv_cached)
        for (int i = lane; i < a.B; i += 64) nv_lds[i] = (uint32_t)valid_len(a, i);
    __builtin_amdgcn_wave_barrier();

    struct Batch {
        uint32_t v[E];        // votes (loaded) -> bin address A (pass 1) -> key (pass 2)
        int64_t cell, rowi;   // cell index; row of answers / tokens it reads (the problem's, for prefix budgets over a pool)
        int32_t b, truth;
        uint32_t n;           // valid votes of this lane's cell (0 when the slot is past the last cell)
        uint32_t sh;          // !VEC: elements between the aligned start of the superset and the row (0..3); VEC: 0
    };
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t stride = nwaves * C;                               // cells between consecutive batches of this wave
    int64_t ncell = wave * C + sub;                                  // walker state of the NEXT batch to load
    int64_t np = ncell / a.B;
    int32_t nb = (int32_t)(ncell - np * a.B);
    const int64_t dp = stride / a.B;
    const int32_t db = (int32_t)(stride - dp * a.B);

    auto element = [&](int k, int j) -> uint32_t {                  // slot (k, j) of this lane inside its cell's (superset) row
        return (uint32_t)((k * G + l) * 4 + j);
    };
    // Loads are UNCONDITIONAL (a vote past the valid prefix re-reads element 0 of the row; a slot past the last
    // cell reads cell 0): the number of loads per batch is then a compile-time constant, so the compiler can
    // wait for the current batch with s_waitcnt vmcnt(<loads of the next batch>) instead of vmcnt(0) --
    // with predicated loads it cannot, and the prefetch would be drained before every batch.
    auto load = [&](Batch& t) {                                     // issues loads only
        t.cell = ncell; t.b = nb;
        const bool live = ncell < a.ncells;
        const int32_t bb = live ? nb : 0;
        t.n = live ? (nv_cached ? nv_lds[bb] : (uint32_t)valid_len(a, bb)) : 0u;
        t.truth = a.truth[live ? np : 0];
        t.rowi = live ? (a.pool_rows ? np : ncell) : 0;
        const int32_t* row = a.answers + t.rowi * a.N;
        t.sh = VEC ? 0u : ((uint32_t)(uintptr_t)row & 15u) >> 2;
        const int4* row4 = reinterpret_cast<const int4*>(row - t.sh);          // 16-byte aligned (a vector never crosses a page)
        // a lane whose vector lies beyond the row re-reads the row's LAST vector: in a row that ends inside a wave instruction (N = 96 in
        // the 128-slot shape) that is the 16 bytes a neighbouring lane reads in the SAME instruction -- one request, no second fetch
        // (round 5 re-read vector 0 there: a non-temporal line is not kept, FETCH_SIZE was 1.36x the row bytes at N = 96)
        const uint32_t vlast = t.n + t.sh ? (t.n + t.sh - 1u) >> 2 : 0u;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const uint32_t vi = (uint32_t)(k * G + l) < vlast ? (uint32_t)(k * G + l) : vlast;
            const int4 x = stream_load(row4 + vi);
            t.v[4 * k] = (uint32_t)x.x; t.v[4 * k + 1] = (uint32_t)x.y; t.v[4 * k + 2] = (uint32_t)x.z; t.v[4 * k + 3] = (uint32_t)x.w;
        }
        ncell += stride; np += dp; nb += db;
        if (nb >= a.B) { nb -= a.B; np += 1; }
    };

    uint32_t bad = 0;
    // pass 1 (o1.py:181-195): h[bin][copy] += 1; inactive votes feed the trash bin.  FULL: every lane's cell has
    // exactly CAP valid votes (wave-uniform), so no vote needs an activity predicate.
    // EL = 4 * (live vectors): slots at or beyond the longest cell of the batch are skipped by every pass (a cell shorter
    // than the shape's capacity, ragged n_valid) -- the cost of a batch follows its votes, not the shape's capacity.
    // FULL: every slot of every lane is a vote (wave-uniform).  Otherwise the slots that are NOT votes -- beyond a cell's valid
    // prefix, or (unaligned rows) in front of the row -- are first replaced by a per-lane sentinel value whose bin address is the
    // lane's trash (SENT below), vector by vector and only in the vectors that can hold such slots: every lane's vectors
    // [first unaligned ? 1 : 0, kfull) are all votes (kfull: wave-uniform, from the shortest cell of the batch).  After that the
    // passes run unmasked: a cell of 1001 votes costs what its 1001 votes cost, not 1024 masked ones.
    auto vote_pass = [&](Batch& c, auto full_tag, auto live_tag, int kfull) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int EL = 4 * decltype(live_tag)::value;
        // activity of vote i as a MASK, not a predicate (64 live SGPR pairs would spill): element(i) < n  <=>
        // const_i < nl with the lane term moved to the right-hand side; m = all ones when active
        const int32_t nl = (int32_t)(c.n + c.sh) - 4 * l;
        // PIVOTS (see the kernel comment): this lane's first (and second) vote.  A vote equal to a pivot is added to a word of
        // the lane's own (TW / TW2: conflict-free) instead of its -- shared, contended -- histogram bin; after the last vote
        // the word is moved into the pivot's bin: one read-and-clear + one add per lane and pivot instead of one add per vote.
        // Two pivots per lane: exact 2-way ties (and a wrong majority next to the truth) are as cheap as one peaked value.
        // (One code path: selecting a one-pivot variant per batch cost 20 VGPRs -- spills at 16 waves -- for 2 VALU per vote.)
        // Domain check of the whole batch up front (o1.py:140 int(extracted_answer) is unbounded; the extractor maps it into
        // bins 0..1023): the per-vote clamp runs only in the -- wave-uniform, rare -- case that some slot holds a larger value.
        {
            uint32_t orv = 0;
#pragma unroll
            for (int i = 0; i < EL; ++i) orv |= c.v[i];
            if (__any(orv > 1023u)) {
#pragma unroll
                for (int i = 0; i < EL; ++i) {
                    const uint32_t v = c.v[i];
                    uint32_t m = 0xffffffffu;
                    if (!FULL) {
                        m = (uint32_t)(((i >> 2) * G * 4 + (i & 3) - nl) >> 31);
                        if (!VEC && i < 3) m &= ~(uint32_t)((int32_t)(4 * l + i - (int32_t)c.sh) >> 31);
                    }
                    bad |= v & m;                                   // (an inactive slot holds a re-read of vector 0 / the row's neighbours)
                    c.v[i] = v < 1023u ? v : 1023u;
                }
            }
        }
        if (!FULL) {
            // SENT: the (negative) "vote" whose bin address KB - (SENT << S) is ATR, the lane's trash.  KB - ATR is a multiple of the
            // bin stride in every 16-bit layout (regions are 8-byte multiples).
            const uint32_t SENT = (uint32_t)((int32_t)(KB - ATR) >> S);
#pragma unroll
            for (int k = 0; k < EL / 4; ++k) {
                if (k < kfull && (VEC || k > 0)) continue;          // (wave-uniform) every slot of this vector is a vote in every lane
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * k + j;
                    bool ok = k * G * 4 + j < nl;
                    if (!VEC && i < 3) ok = ok && (4 * l + i >= (int32_t)c.sh);     // the <= 3 slots before the row
                    c.v[i] = ok ? c.v[i] : SENT;
                }
            }
        }
        // bin addresses of all votes (a sentinel's: the lane's trash address), then the pivots: the lane's first vote, and the
        // first of its next three votes that differs from it
        uint32_t A[EL];
#pragma unroll
        for (int i = 0; i < EL; ++i) {
            A[i] = bin_address<S>(KB, c.v[i]);
            c.v[i] = A[i];                                          // (a pivot vote keeps its bin address: the bin holds the total when it is read)
        }
        const uint32_t ap0 = A[0];                                  // an inactive lane's pivot is its trash address
        uint32_t ap1 = A[1] != ap0 ? A[1] : (A[2] != ap0 ? A[2] : A[3]);
#pragma unroll
        for (int i = 0; i < EL; ++i) {
            if (B8) {
                uint32_t inc, W;
#ifdef SCV_G8_NOPIVOT
                W = A[i] & ~3u; inc = byte_inc(A[i]);               // (A/B builds: every vote straight to its bin)
#else
                if (i == 0) { W = TW; inc = byte_inc(A[0]); }
                else W = pivot_select_b8(A[i], ap0, ap1, TW, TW2, inc);
#endif
                lds_add(W, inc);
            } else if (H16) {
                uint32_t inc, W;
                if (i == 0) { W = TW; inc = __builtin_amdgcn_alignbyte(1u, 1u, A[0]); }
                else W = pivot_select_h16(A[i], ap0, ap1, TW, TW2, inc);
                lds_add(W, inc);
            } else {
                const uint32_t W = i == 0 ? TW : pivot_select(A[i], copy4, ap0, ap1, TW, TW2);
                lds_add(W, copy_inc);
            }
        }
        // move the pivot words into the pivots' bins (LDS operations of a wave execute in order: the reads of pass 2 see them)
#ifdef SCV_G8_NOPIVOT
        if (!B8)
#endif
        {
            const uint32_t w0 = __hip_atomic_exchange(reinterpret_cast<lds_u32*>((uintptr_t)TW), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t w1 = __hip_atomic_exchange(reinterpret_cast<lds_u32*>((uintptr_t)TW2), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            lds_add(H16 ? (ap0 + copy4) & ~3u : (ap0 | copy4), w0);
            // (a second pivot that collected nothing -- it equals the first, or is off -- adds its zero to the lane's trash: a zero
            //  added to a hot bin still queues behind the other lanes' adds to that word)
            const uint32_t t1 = w1 ? ap1 : ATR;
            lds_add(H16 ? (t1 + copy4) & ~3u : (t1 | copy4), w1);
        }
        // the lane's trash has collected the non-votes (and empty pivots): zero it, so that pass 2 reads count 0 there and needs no masks
        if (!FULL) *reinterpret_cast<lds_v2u*>((uintptr_t)ATR) = scv_v2u{0u, 0u};
        __builtin_amdgcn_wave_barrier();
    };
    // passes 2-4, sparse: read back the counts of the bins this lane voted for
    auto sparse_passes = [&](Batch& c, auto live_tag, uint32_t& gkey, uint32_t& at_max, uint32_t& tc) {
        constexpr int EL = 4 * decltype(live_tag)::value;
        uint32_t lmax = 0;
        constexpr int CH = EL > 16 ? 8 : EL;                          // bound the registers held by reads in flight
        // h[truth] (o1.py:206): every lane of the cell reads the truth's bin (same address: a broadcast).  Issued first, so it
        // has returned when the counts of pass 2 have (LDS answers in order); pinned before the clears of pass 4 below.
        const uint32_t ATt = (c.truth >= 0 && c.truth < kBins) ? KB - ((uint32_t)c.truth << S) : ATR;   // no such bin: the lane's trash
        tc = B8 ? (uint32_t)*reinterpret_cast<lds_u8*>((uintptr_t)ATt) : (H16 ? lds_count16<R>(ATt) : lds_count_packed<R>(ATt));
#pragma unroll
        for (int i0 = 0; i0 < EL; i0 += CH) {
            // all CH reads are issued before the first count is consumed (left alone, the scheduler keeps only
            // two ds_reads in flight and the pass becomes a chain of LDS latencies: measured 48 % wave-wait)
            uint32_t cn[CH];
#pragma unroll
            for (int i = i0; i < i0 + CH; ++i) cn[i - i0] = B8 ? (uint32_t)*reinterpret_cast<lds_u8*>((uintptr_t)c.v[i]) : (H16 ? lds_count16<R>(c.v[i]) : lds_count_packed<R>(c.v[i]));
            __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, CH, 0);
            __builtin_amdgcn_sched_group_barrier(0x2 /* VALU */, CH * 6, 0);
#pragma unroll
            for (int i = i0; i < i0 + CH; ++i) {
                const uint32_t A = c.v[i];
                uint32_t key = (cn[i - i0] << kKeyShift) | A;
                c.v[i] = key;                                           // (a non-vote reads its lane's trash, zeroed after pass 1: count 0, below
                                                                        //  every real key; pass 4 then clears the trash again, never another region)
                lmax = key > lmax ? key : lmax;
            }
        }
        gkey = cellgroup_max<G>(lmax);                              // every lane of the cell gets it
        const uint32_t thr = gkey & ~kKeyMask;                       // max_count << kKeyShift
        // pass 3 (statistics.py:599-601): votes at max -> number of distinct modes
        at_max = 0;
        // The truth's count is consumed here: the compiler may not sink its load below the 16-bit clears that follow (the load
        // is a 32- / 64-bit access: type-based alias analysis would let it).
        asm volatile("" : "+v"(tc) : : "memory");
        if (ATt == ATR) tc = 0u;
        // ... and pass 4 in the same loop: sparse clear (an inactive vote's key addresses the lane's trash).  LDS
        // operations of a wave execute in order, so the clears need no wait for the reads of pass 2; interleaved, the
        // address arithmetic of the clear fills the wait state between v_cmp and the carry-in add of the count.
#pragma unroll
        for (int i = 0; i < EL; ++i) {
            at_max += c.v[i] >= thr ? 1u : 0u;                      // inactive keys have count 0: they only count when max_count == 0
            if (B8) *reinterpret_cast<lds_u8*>((uintptr_t)(c.v[i] & kKeyMask)) = (uint8_t)0;
            else if (H16) *reinterpret_cast<lds_u16*>((uintptr_t)((c.v[i] & kKeyMask) + copy4)) = (uint16_t)0;
            else *reinterpret_cast<lds_u16*>((uintptr_t)((c.v[i] & kKeyMask) | copy2)) = (uint16_t)0;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // The outcome of one batch for this lane's cell: everything the record and the counters need, the same in every lane of the cell.
    struct Outcome { uint32_t gkey, sum_at_max, tc; long long tok; int64_t cell; int32_t b; };
    auto reduce_cell = [&](const Batch& c, uint32_t gkey, uint32_t at_max, uint32_t tc, long long tsum) -> Outcome {
        Outcome o;
        o.gkey = gkey;
        o.sum_at_max = cellgroup_sum<G>(at_max);                             // votes at the maximum, over the cell
        o.tc = tc;
        o.tok = 0;
        if (TOK) o.tok = cellgroup_sum_i64<G>(tsum);
        o.cell = c.cell; o.b = c.b;
        return o;
    };
    // statistics.multimode + o1.py:202-206: the record and the counters of one cell, by ONE lane of the cell
    auto emit_cell = [&](const Outcome& o) {
        const uint32_t maxc = o.gkey >> kKeyShift;                          // max count over every value of the cell
        const bool any = maxc > 0;
        const uint32_t n_modes = any ? exact_quotient(o.sum_at_max, maxc) : 0u;
        const uint32_t mm = 1023u - (((o.gkey & kKeyMask) - cellbase) >> S);
        const uint32_t hit = (any && o.tc == maxc) ? 1u : 0u;               // o1.py:206
        if (a.cells) {
            uint4 rec;
            rec.x = maxc;
            rec.y = o.tc;
            rec.z = (n_modes & 0xffffu) | ((any ? (mm & 0xffffu) : 0xffffu) << 16);
            rec.w = hit;
            if (a.packed_cells) reinterpret_cast<uint32_t*>(a.cells)[o.cell] = pack_cell(maxc, o.tc, n_modes, mm, hit);
            else reinterpret_cast<uint4*>(a.cells)[o.cell] = rec;
        }
        if (a.cell_tokens) a.cell_tokens[o.cell] = o.tok;
        if (wgc.tcl > 0) wg_counters_add<TOK>(a, wgc, o.b, hit, n_modes, o.tc, o.tok);
        else {
            if (a.tie_hits && hit) atomicAdd(&a.tie_hits[(int64_t)o.b * SCV_TIE_CLASSES + n_modes], 1ull);
            if (TOK && a.token_sum) atomicAdd(&a.token_sum[o.b], (unsigned long long)o.tok);
            if (a.truth_sum) atomicAdd(&a.truth_sum[o.b], (unsigned long long)o.tc);
        }
    };

    // one loop step: count the batches in `cur` while the batches of `nxt` are in flight
    auto step = [&](Batch (&cur)[K], Batch (&nxt)[K], bool more) {
        if (more) {
#pragma unroll
            for (int j = 0; j < K; ++j) load(nxt[j]);
        }
        // (with tokens the K outcomes held across the step cost spills at the 128-register cap of the K >= 2 shapes: one block per batch there)
        constexpr bool kOneMask = K > 1 && !TOK;
        Outcome out[kOneMask ? K : 1];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            Batch& c = cur[j];
            const uint32_t n = c.n;
            // tokens of the current batch: issued now, consumed after the LDS passes (only the sum is needed)
            long long tsum = 0;
            if (TOK) {
                // the token row has its own alignment (its base may differ from the votes'): same superset trick, only the sum is needed
                const int32_t* trow = a.tokens + c.rowi * a.N;
                const uint32_t tsh = VEC ? 0u : ((uint32_t)(uintptr_t)trow & 15u) >> 2;
                const int4* trow4 = reinterpret_cast<const int4*>(trow - tsh);
                const uint32_t hi = n + tsh;
                const uint32_t tlast = hi ? (hi - 1u) >> 2 : 0u;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const uint32_t vi = (uint32_t)(k * G + l) < tlast ? (uint32_t)(k * G + l) : tlast;
                    const int4 y = stream_load(trow4 + vi);
                    // slot e is token e - tsh: valid iff tsh <= e < n + tsh (unsigned: e - tsh < n)
                    tsum += (element(k, 0) - tsh < n ? (long long)y.x : 0) + (element(k, 1) - tsh < n ? (long long)y.y : 0)
                          + (element(k, 2) - tsh < n ? (long long)y.z : 0) + (element(k, 3) - tsh < n ? (long long)y.w : 0);
                }
            }
            uint32_t gkey, at_max, tc;
            int kfull = 0;
            auto partial = [&](auto live_tag) {
                vote_pass(c, std::false_type{}, live_tag, kfull);
                sparse_passes(c, live_tag, gkey, at_max, tc);
            };
            if (__all(n == CAP && (VEC || c.sh == 0u))) {
                vote_pass(c, std::true_type{}, std::integral_constant<int, V>{}, V);
                sparse_passes(c, std::integral_constant<int, V>{}, gkey, at_max, tc);
            } else {
                // longest and shortest cell of the batch (wave-uniform: lane s * G holds cell slot s) -> live vectors, and the
                // vectors [0, kfull) whose every slot is a vote in every lane
                const uint32_t nsh = n + c.sh;                     // slots of the (superset) row in use
                uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)nsh, 0), nmin = nmax;
#pragma unroll
                for (int sl = 1; sl < C; ++sl) {
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)nsh, sl * G);
                    nmax = o > nmax ? o : nmax;
                    nmin = o < nmin ? o : nmin;
                }
                kfull = (int)(nmin / (4u * G));
                const uint32_t live = (nmax + 4u * G - 1u) / (4u * G);
                if (V >= 4 && live > 3) partial(std::integral_constant<int, V >= 4 ? 4 : V>{});
                else if (V >= 3 && live > 2) partial(std::integral_constant<int, V >= 3 ? 3 : V>{});
                else if (V >= 2 && live > 1) partial(std::integral_constant<int, V >= 2 ? 2 : V>{});
                else partial(std::integral_constant<int, 1>{});
            }
            out[kOneMask ? j : 0] = reduce_cell(c, gkey, at_max, tc, tsum);
            if (!kOneMask && l == 0 && c.cell < a.ncells) emit_cell(out[0]);
        }
        if (!kOneMask) return;
        // The records and counters of the step's K batches under ONE mask: lane j of a cell takes batch j (every lane of a cell holds its
        // cell's outcome of every batch).  With one masked block per batch the ~40 wave-instructions of the block -- quotient, record, LDS
        // counters, all for 64 / G active lanes -- were issued K times per step (N <= 128: K = 2 or 4, 8 % of the step at N = 128).
        Outcome mine = out[0];
#pragma unroll
        for (int j = 1; j < K; ++j)
            if (l == j) mine = out[j];
        if (l < K && mine.cell < a.ncells) emit_cell(mine);
    };

    Batch bufa[K], bufb[K];
    const int64_t nbatches = (a.ncells + C - 1) / C;
    const int64_t istride = nwaves * K;
    // iteration `it` of this wave covers batches it + j * nwaves, j < K (slots past the end count nothing);
    // two buffers ping-pong so that no register copies are needed
#pragma unroll
    for (int j = 0; j < K; ++j) load(bufa[j]);
    for (int64_t it = wave; it < nbatches; it += 2 * istride) {
        step(bufa, bufb, it + istride < nbatches);
        if (it + istride >= nbatches) break;
        step(bufb, bufa, it + 2 * istride < nbatches);
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    wg_counters_flush<TOK>(a, wgc, (int)threadIdx.x, (int)blockDim.x);
}

// ---- kernel 1h: register-streamed long cells (1024 < N <= 4096 and up), dense scan ----------------
//
// Same building blocks as scv_reg_cells with G = 64 (one wave per cell, R = 4 copies, no barrier), but a cell
// is streamed in H parts of V vectors per lane (16-bit counters, the 4 copies of a bin packed in 8 bytes: 8 KiB of
// LDS per wave): part s+1 is in flight while part s is voted, and since the
// dense scan never looks at the votes again only 2 x 4V registers hold votes whatever the cell length
// (N = 4096: 64 registers instead of 128 -> 2+ waves per SIMD instead of 1).  After the last part every lane
// scans its 16 bins (8 ds_read_b128 at immediate offsets, v_dot2_u32_u16 sums the copies), zeroes them, and the wave reduces.
template <int V, int H, bool TOK, bool VEC>
__global__ __launch_bounds__((64 * reg_dense_waves<V, H, TOK, VEC>())) void scv_reg_dense(const AggArgs a) {
    constexpr int S = 3;                      // 8 bytes per bin: 4 copies of a 16-bit counter (a cell holds <= 65535 votes here)
    constexpr int E = 4 * V;                  // votes per lane per part
    constexpr uint32_t PART = 256u * V;       // votes per part
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_wg[];
    const int lane = threadIdx.x & 63;
    uint32_t* smem = smem_wg + (threadIdx.x >> 6) * a.wave_lds_words;   // independent waves, grouped for SIMD balance
    const uint32_t base = (uint32_t)(uintptr_t)(lds_u32*)smem;
    {
        uint4* h4 = reinterpret_cast<uint4*>(smem);
        for (int i = lane; i < kRegWaveWords16 / 4; i += 64) h4[i] = make_uint4(0, 0, 0, 0);
    }
    const WgCounters wgc = wg_counters_begin(a, smem_wg + (blockDim.x >> 6) * a.wave_lds_words, (int)threadIdx.x, (int)blockDim.x);
    const uint32_t KB = base + (1023u << S);
    const uint32_t ATR = base + (uint32_t)(kRegHist16Words + 64) * 4u + (uint32_t)lane * 8u;   // this lane's trash
    const uint32_t copy4 = (((uint32_t)lane & 3u) >> 1) * 4u;             // word of this lane's copy inside a bin
    const uint32_t copy_inc = 1u << (16u * ((uint32_t)lane & 1u));        // ... and its half of that word
    const uint32_t A0 = base + (uint32_t)lane * 16u;                      // scan: 16 bytes = bins (2 x, 2 x + 1), x = lane + 64 j
    const uint32_t TW = base + (uint32_t)kRegHist16Words * 4u + (uint32_t)lane * 4u;   // this lane's word of the wide truth bin
    uint32_t* nv_lds = smem + kRegWaveWords16;
    const bool nv_cached = a.n_valid && a.B <= kMaxSortedB;
    if (nv_cached)
        for (int i = lane; i < a.B; i += 64) nv_lds[i] = (uint32_t)valid_len(a, i);
    __builtin_amdgcn_wave_barrier();

    struct Part {
        uint32_t v[E];
        int32_t tk[TOK ? E : 1];
        uint32_t nrel;        // slots of the cell's (superset) row in use, minus the slots before this part (may be <= 0 as int)
        uint32_t sh;          // !VEC: elements between the aligned start of the superset and the row (0..3); VEC: 0
        uint32_t tnrel, tsh;  // !VEC && TOK: the same for the token row (its base may be aligned differently)
    };
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t ncell = wave;                     // cell whose parts are being loaded
    int64_t np = ncell / a.B;
    int32_t nb = (int32_t)(ncell - np * a.B);
    const int64_t dp = nwaves / a.B;
    const int32_t db = (int32_t)(nwaves - dp * a.B);
    uint32_t ln = 0;                          // valid_len of the cell being loaded
    int lpart = 0, lneed = 1;                 // part being loaded, parts this cell needs
    // A cell of n votes is streamed in ceil(slots / PART) parts, not in H (wave-uniform: one cell per wave): N = 4501 on the 8-part
    // shape costs 5 parts, a budget of 7 votes one.  (slots: the row's aligned superset holds up to 3 more than n; at least one part,
    // which also carries the cell's pivots.)
    auto parts_needed = [&](uint32_t n) -> int {
        const uint32_t slots = n + (VEC ? 0u : 3u);
        const int need = (int)((slots + PART - 1u) / PART);
        return need < 1 ? 1 : (need > H ? H : need);
    };
    auto begin_cell_load = [&]() {
        const bool live = ncell < a.ncells;
        const int32_t bb = live ? nb : 0;
        ln = live ? (nv_cached ? nv_lds[bb] : (uint32_t)valid_len(a, bb)) : 0u;
        lneed = parts_needed(ln);
    };
    // unconditional loads (constant count per part -> counted vmcnt waits), clamped to vector 0 of the row.  !VEC: the row's
    // 16-byte-aligned superset is read (see scv_reg_cells): slot e of the superset is vote e - sh.
    auto load_part = [&](Part& t, auto dyn_tag) __attribute__((always_inline)) {
        constexpr bool DYN = decltype(dyn_tag)::value;        // parts per cell from its length (else: every cell takes H parts)
        const bool live = ncell < a.ncells;
        const int64_t rowoff = (live ? (a.pool_rows ? np : ncell) : 0) * a.N;
        const int32_t* row = a.answers + rowoff;
        t.sh = VEC ? 0u : ((uint32_t)(uintptr_t)row & 15u) >> 2;
        const int4* row4 = reinterpret_cast<const int4*>(row - t.sh);
        const int32_t nrel = (int32_t)(ln + t.sh) - (int32_t)(lpart * PART);
        t.nrel = (uint32_t)nrel;
        const int4* trow4 = nullptr;
        int32_t tnrel = nrel;
        t.tsh = t.sh;
        if (TOK) {
            const int32_t* trow = a.tokens + rowoff;
            t.tsh = VEC ? 0u : ((uint32_t)(uintptr_t)trow & 15u) >> 2;
            trow4 = reinterpret_cast<const int4*>(trow - t.tsh);
            tnrel = (int32_t)(ln + t.tsh) - (int32_t)(lpart * PART);
        }
        t.tnrel = (uint32_t)tnrel;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int32_t e0 = (k * 64 + lane) * 4;                  // first slot of this lane's vector inside the part
            const int4 x = stream_load(e0 < nrel ? row4 + (int64_t)lpart * (PART / 4) + (k * 64 + lane) : row4);
            t.v[4 * k] = (uint32_t)x.x; t.v[4 * k + 1] = (uint32_t)x.y; t.v[4 * k + 2] = (uint32_t)x.z; t.v[4 * k + 3] = (uint32_t)x.w;
            if (TOK) {
                const int4 y = stream_load(e0 < tnrel ? trow4 + (int64_t)lpart * (PART / 4) + (k * 64 + lane) : trow4);
                t.tk[4 * k] = y.x; t.tk[4 * k + 1] = y.y; t.tk[4 * k + 2] = y.z; t.tk[4 * k + 3] = y.w;
            }
        }
        if (++lpart == (DYN ? lneed : H)) {
            lpart = 0;
            ncell += nwaves; np += dp; nb += db;
            if (nb >= a.B) { nb -= a.B; np += 1; }
            begin_cell_load();
        }
    };

    uint32_t bad = 0;
    long long tsum = 0;
    // PIVOTS (as in scv_reg_cells): the lane's first (and, when the cell shows two hot values, second) vote of the CELL; votes
    // equal to a pivot go to the lane's own words TW / TW2 across all parts and join their bins once, before the scan.
    uint32_t ap0 = 0xffffffffu, ap1 = 0xffffffffu;
    const uint32_t TW2 = base + (uint32_t)(kRegHist16Words + 192) * 4u + (uint32_t)lane * 4u;
    auto vote_part = [&](const Part& c, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;         // first part of a cell: slots 0 (and 1) set the pivots
        const int32_t nl = (int32_t)c.nrel - 4 * lane;
        const int32_t tnl = (int32_t)c.tnrel - 4 * lane;              // (tokens: their own window when !VEC)
        auto one = [&](int i, uint32_t A) {
            lds_add(pivot_select(A, copy4, ap0, ap1, TW, TW2), copy_inc);
        };
        if (FIRST) {
            // the cell's pivots: the lane's first vote and the first of its next three that differs (an inactive slot: the trash address)
            uint32_t A4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t m = (uint32_t)(((i & 3) - nl) >> 31) & ~(VEC ? 0u : (uint32_t)((int32_t)(4 * lane + i - (int32_t)c.sh) >> 31));
                const uint32_t v = c.v[i];
                A4[i] = (bin_address<S>(KB, v < 1023u ? v : 1023u) & m) | (ATR & ~m);
            }
            ap0 = A4[0];
            ap1 = A4[1] != ap0 ? A4[1] : (A4[2] != ap0 ? A4[2] : A4[3]);
            }
        // a part is FULL when every slot of it is a vote (and a token): no masks
        bool full = (int32_t)c.nrel >= (int32_t)PART;
        if (!VEC) full = full && (!TOK || (int32_t)c.tnrel >= (int32_t)PART) && (!FIRST || (c.sh == 0u && c.tsh == 0u));
        // domain check of the whole part up front: the per-vote clamp only runs in the (wave-uniform, rare) case that some slot
        // -- a vote, or a neighbour of the row that is masked out below -- holds a value above 1023
        uint32_t orv = 0;
#pragma unroll
        for (int i = 0; i < E; ++i) orv |= c.v[i];
        const bool clamp = __any(orv > 1023u);
        if (__all(full)) {
            if (clamp) bad |= orv;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const uint32_t v = c.v[i];
                one(i, bin_address<S>(KB, clamp ? (v < 1023u ? v : 1023u) : v));
                if (TOK) tsum += c.tk[i];
            }
        } else {
            // A part that is not full: vector by vector (one cell per wave, so every test is wave-uniform).  Vectors past the end
            // of the cell are skipped, vectors whose 256 slots are all votes (and tokens) run unmasked -- only the vector that
            // holds the end of the row (and, for unaligned rows, the first vector of the cell) pays for masks: a cell of 1000
            // votes costs what its votes cost, not 1024 masked ones (N = 1000 on the one-part shape: 210 -> 19x us).
            const int32_t beyond = (TOK && !VEC && (int32_t)c.tnrel > (int32_t)c.nrel) ? (int32_t)c.tnrel : (int32_t)c.nrel;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                if (k * 256 >= beyond) continue;
                bool vfull = (int32_t)c.nrel >= (k + 1) * 256 && (!TOK || VEC || (int32_t)c.tnrel >= (k + 1) * 256);
                if (!VEC && FIRST && k == 0) vfull = vfull && c.sh == 0u && c.tsh == 0u;
                if (vfull) {
                    if (clamp) bad |= c.v[4 * k] | c.v[4 * k + 1] | c.v[4 * k + 2] | c.v[4 * k + 3];
#pragma unroll
                    for (int i = 4 * k; i < 4 * k + 4; ++i) {
                        const uint32_t v = c.v[i];
                        one(i, bin_address<S>(KB, clamp ? (v < 1023u ? v : 1023u) : v));
                        if (TOK) tsum += c.tk[i];
                    }
                    continue;
                }
#pragma unroll
                for (int i = 4 * k; i < 4 * k + 4; ++i) {
                    // (the pivot slots of a first part must define the pivots: an inactive one becomes the lane's trash address)
                    const uint32_t v = c.v[i];
                    const int32_t ci = (i >> 2) * 256 + (i & 3);
                    uint32_t m = (uint32_t)((ci - nl) >> 31);                       // all ones when the vote is valid
                    if (!VEC && FIRST && i < 3) m &= ~(uint32_t)((int32_t)(4 * lane + i - (int32_t)c.sh) >> 31);   // the <= 3 slots before the row
                    if (clamp) bad |= v & m;
                    uint32_t A = bin_address<S>(KB, clamp ? (v < 1023u ? v : 1023u) : v);
                    A = (A & m) | (ATR & ~m);
                    one(i, A);
                    if (TOK) {
                        uint32_t mt = m;
                        if (!VEC) {
                            mt = (uint32_t)((ci - tnl) >> 31);
                            if (FIRST && i < 3) mt &= ~(uint32_t)((int32_t)(4 * lane + i - (int32_t)c.tsh) >> 31);
                        }
                        tsum += (long long)(c.tk[i] & (int32_t)mt);
                    }
                }
            }
        }
    };
    auto finish_cell = [&](int64_t cell, int32_t b, int32_t truth) {
        // the pivot words join the pivots' bins (read-and-clear, then one add per lane and pivot)
        {
            const uint32_t w0 = __hip_atomic_exchange(reinterpret_cast<lds_u32*>((uintptr_t)TW), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            lds_add(ap0 | copy4, w0);
            const uint32_t w1 = __hip_atomic_exchange(reinterpret_cast<lds_u32*>((uintptr_t)TW2), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            lds_add((w1 ? ap1 : ATR) | copy4, w1);                     // (a pivot that collected nothing adds its zero to the lane's trash)
        }
        // h[truth] (o1.py:206): the four 16-bit copies of the truth's bin, read before the scan zeroes them
        uint32_t tc = 0;
        if (truth >= 0 && truth < kBins) tc = lds_count_packed<4>(KB - ((uint32_t)truth << S));
        __builtin_amdgcn_wave_barrier();
        uint32_t key[16];
        uint32_t lmax = 0;
        {
            scv_v4u q[8];                                           // 8 b128 reads (32 registers) in flight
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<lds_v4u*>((uintptr_t)(A0 + 1024u * j));
            __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, 8, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                key[2 * j] = (sum_halves(q[j].x, sum_halves(q[j].y, 0u)) << kKeyShift) | (A0 + 1024u * j);
                key[2 * j + 1] = (sum_halves(q[j].z, sum_halves(q[j].w, 0u)) << kKeyShift) | (A0 + 1024u * j + 8u);
                lmax = key[2 * j] > lmax ? key[2 * j] : lmax;
                lmax = key[2 * j + 1] > lmax ? key[2 * j + 1] : lmax;
            }
        }
        asm volatile("" : "+v"(tc) : : "memory");                    // h[truth] has returned with the scan's reads: pinned before the zeroing stores
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<lds_v4u*>((uintptr_t)(A0 + 1024u * j)) = scv_v4u{0u, 0u, 0u, 0u};
        __builtin_amdgcn_wave_barrier();
        const uint32_t gkey = cellgroup_max<64>(lmax);
        const uint32_t thr = gkey & ~kKeyMask;
        uint32_t at_max = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) at_max += key[j] >= thr ? 1u : 0u;    // BINS at max (<= 16 per lane)
        const uint32_t bins_at_max = cellgroup_sum<64>(at_max);
        long long tok = 0;
        if (TOK) tok = cellgroup_sum_i64<64>(tsum);
        tsum = 0;
        if (lane == 0) {
            // statistics.multimode + o1.py:202-206
            const uint32_t maxc = gkey >> kKeyShift;
            const bool any = maxc > 0;
            const uint32_t mm = 1023u - (((gkey & kKeyMask) - base) >> S);
            const uint32_t hit = (any && tc == maxc) ? 1u : 0u;             // o1.py:206
            const uint32_t n_modes = any ? bins_at_max : 0u;
            if (a.cells) {
                uint4 rec;
                rec.x = maxc;
                rec.y = tc;
                rec.z = (n_modes & 0xffffu) | ((any ? (mm & 0xffffu) : 0xffffu) << 16);
                rec.w = hit;
                reinterpret_cast<uint4*>(a.cells)[cell] = rec;
            }
            if (a.cell_tokens) a.cell_tokens[cell] = tok;
            if (wgc.tcl > 0) wg_counters_add<TOK>(a, wgc, b, hit, n_modes, tc, tok);
            else {
                if (a.tie_hits && hit) atomicAdd(&a.tie_hits[(int64_t)b * SCV_TIE_CLASSES + n_modes], 1ull);
                if (TOK && a.token_sum) atomicAdd(&a.token_sum[b], (unsigned long long)tok);
                if (a.truth_sum) atomicAdd(&a.truth_sum[b], (unsigned long long)tc);
            }
        }
    };

    // stream of parts: the cells of this wave one after the other, each in as many parts as it needs; two buffers ping-pong.  The
    // loading side runs one part ahead of the counting side over the same sequence, so the loop ends when the last cell is counted.
    int64_t ccell = wave;                     // cell being counted
    int64_t cp = np;
    int32_t cb = nb;
    int cpart = 0;
    auto cell_len = [&](int32_t b) -> uint32_t { return nv_cached ? nv_lds[b] : (uint32_t)valid_len(a, b); };
    int cneed = ccell < a.ncells ? parts_needed(cell_len(cb)) : 1;
    int32_t ctruth = ccell < a.ncells ? a.truth[cp] : 0;
    auto count_part = [&](const Part& c, auto dyn_tag) __attribute__((always_inline)) {
        constexpr bool DYN = decltype(dyn_tag)::value;
        if (cpart == 0) vote_part(c, std::true_type{});
        else vote_part(c, std::false_type{});
        if (++cpart == (DYN ? cneed : H)) {
            cpart = 0;
            finish_cell(ccell, cb, ctruth);
            ccell += nwaves; cp += dp; cb += db;
            if (cb >= a.B) { cb -= a.B; cp += 1; }
            if (ccell < a.ncells) { ctruth = a.truth[cp]; if (DYN) cneed = parts_needed(cell_len(cb)); }
        }
    };
    Part pa, pb;
    begin_cell_load();
    // Every cell full length (no n_valid, N needs all H parts): the part count is the compile-time H, and with H even the two
    // buffers keep their roles (first part / last part) -- the dynamic form costs such launches 5-8 %.
#define SCV_DENSE_STREAM(TAG)                                                     \
    do {                                                                          \
        if (ncell < a.ncells) load_part(pa, TAG);                                 \
        while (ccell < a.ncells) {                                                \
            if (ncell < a.ncells) load_part(pb, TAG);                             \
            count_part(pa, TAG);                                                  \
            if (ccell >= a.ncells) break;                                         \
            if (ncell < a.ncells) load_part(pa, TAG);                             \
            count_part(pb, TAG);                                                  \
        }                                                                         \
    } while (0)
    if (!a.n_valid && parts_needed((uint32_t)a.N) == H) SCV_DENSE_STREAM(std::false_type{});
    else SCV_DENSE_STREAM(std::true_type{});
#undef SCV_DENSE_STREAM
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    wg_counters_flush<TOK>(a, wgc, (int)threadIdx.x, (int)blockDim.x);
}

// ---- kernel 1e: prefix budgets over one sample pool ---------------------------
//
// The reference's budgets T >= 2^11 vote over PREFIXES of one pool of samples per problem
// (o1.py:274-277: N = T // 2048 samples idx 0..N-1 of the same cache keys, o1.py:85-88).  Instead of a
// dense [P, B, N] tensor that repeats the pool B times, stream pool[p, 0:max n_valid] ONCE and
// snapshot the running histogram at every boundary (ascending n_valid): fold without zeroing ->
// cell (p, b).  Algorithmic bytes: 4 * max_b n_valid[b] per problem instead of 4 * sum_b n_valid[b].
// Here a.answers / a.tokens are [P, N] and a.ncells = P (work items are problems).
template <int RL2, int T, int U, bool TOK>
__global__ __launch_bounds__(T) void scv_prefix_hist(const AggArgs a) {
    constexpr int R = 1 << RL2;
    constexpr int NB = kBins / T;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* hist = smem;
    uint32_t* red = smem + kBins * R;
    int32_t* ord = reinterpret_cast<int32_t*>(red + kRedWords);   // budgets in DESCENDING n_valid order
    const int tid = threadIdx.x;
    const uint32_t copy = (uint32_t)(tid & 63) & (R - 1);
    {
        uint4* h4 = reinterpret_cast<uint4*>(hist);
        for (int i = tid; i < kBins * R / 4; i += T) h4[i] = make_uint4(0, 0, 0, 0);
    }
    build_budget_order(a, ord, tid, T);   // host guarantees n_valid != NULL, B <= kMaxSortedB, sorted = 1
    int32_t* nvb = ord + kMaxSortedB;     // valid length per budget (a load of n_valid per boundary and problem would stall the workgroup each time)
    for (int b = tid; b < a.B; b += T) nvb[b] = (int32_t)valid_len(a, b);
    __syncthreads();

    uint32_t bad = 0;
    for (int64_t p = blockIdx.x; p < a.P; p += gridDim.x) {
        const int32_t* row = a.answers + p * a.N;
        const int32_t* trow = TOK ? a.tokens + p * a.N : nullptr;
        const int32_t truth = a.truth[p];
        long long tsum = 0;          // running token sum of this thread over the prefix so far
        int64_t done = 0;
        for (int32_t k = a.B - 1; k >= 0; --k) {               // ascending n_valid
            const int32_t b = ord[k];
            const int64_t n = nvb[b];
            if (n > done) {
                stream_row<RL2, T, U, TOK>(a, hist, copy, row + done, TOK ? trow + done : nullptr, n - done, tid, bad, tsum);
                done = n;
            }
            if (tid == 0) red[48] = 0;
            __syncthreads();                                    // votes up to n are in LDS
            uint32_t cnt[NB];
            if (k == 0) fold_copies<RL2, T, true>(hist, tid, cnt);   // last (longest) budget: re-arm for the next problem
            else fold_copies<RL2, T, false>(hist, tid, cnt);
            finalize_cell<T, TOK>(a, red, cnt, tsum, tid, p * a.B + b, b, truth);
            // finalize_cell's barriers order this snapshot's reads before the next run's votes
        }
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
}

// ---- synthetic generator ------------------------------------------------------------------------

struct ProblemParams { uint32_t truth, q_num, d[4]; };

__device__ __forceinline__ ProblemParams problem_params(uint64_t seed, int64_t p) {
    ProblemParams r;
    const uint64_t k = mix64((seed ^ 0x5851F42D4C957F2Dull) + kGolden * (uint64_t)(p + 1));
    r.truth = mulhi32((uint32_t)k, 1000u);
    r.q_num = 1u + (uint32_t)(k >> 32) % 7u;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.d[j] = mulhi32((uint32_t)mix64(k + kGolden * (uint64_t)(j + 1)), 1000u);
    return r;
}

#ifdef SCV_TU_MAIN   // non-template kernels: emitted once, by csrc/scvote.hip (the header is shared by several translation units)
__global__ __launch_bounds__(256) void scv_synth_fill_k(int32_t* answers, int32_t* tokens, int32_t* truth,
                                                        int64_t P, int32_t B, int64_t N, int64_t p_offset,
                                                        uint64_t seed, int dist) {
    const int64_t ncells = P * B;
    for (int64_t cell = blockIdx.x; cell < ncells; cell += gridDim.x) {
        const int64_t pl = cell / B;
        const int32_t b = (int32_t)(cell - pl * B);
        const int64_t p = p_offset + pl;
        const ProblemParams pp = problem_params(seed, p);
        if (truth && b == 0 && threadIdx.x == 0) truth[pl] = (int32_t)pp.truth;
        if (!answers && !tokens) continue;
        const uint32_t t0 = pp.q_num * 429496729u;
        const uint32_t T5 = 214748364u;
        const uint32_t m = 2u + (uint32_t)(p & 1);
        const uint32_t base = ((p >> 1) & 1) ? (pp.truth + 500u) % 1000u : pp.truth;
        const int64_t full = (N / m) * m;
        const uint64_t e0 = ((uint64_t)p * (uint64_t)B + (uint64_t)b) * (uint64_t)N;
        int32_t* arow = answers ? answers + cell * N : nullptr;
        int32_t* trow = tokens ? tokens + cell * N : nullptr;
        for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
            const uint64_t u = mix64(seed + kGolden * (e0 + (uint64_t)i + 1));
            if (arow) {
                const uint32_t hi = (uint32_t)(u >> 32), uv = mulhi32((uint32_t)u, 1000u);
                uint32_t v;
                if (dist == SCV_DIST_UNIFORM) v = uv;
                else if (dist == SCV_DIST_PEAKED) {
                    if (hi < t0) v = pp.truth;
                    else {
                        const uint32_t x = hi - t0;
                        const uint32_t j = (x >= T5) + (x >= 2u * T5) + (x >= 3u * T5);
                        v = (x < 4u * T5) ? pp.d[j] : uv;
                    }
                } else if (dist == SCV_DIST_DEGENERATE) v = pp.truth;
                else if (dist == SCV_DIST_PEAKED_WRONG) {
                    const uint32_t hot = pp.d[0] == pp.truth ? (pp.truth + 500u) % 1000u : pp.d[0];
                    if (hi < t0) v = hot;
                    else {
                        const uint32_t x = hi - t0;
                        const uint32_t j = (x >= T5) + (x >= 2u * T5) + (x >= 3u * T5);
                        v = (x < 4u * T5) ? (j == 0 ? pp.truth : pp.d[j]) : uv;
                    }
                } else if (dist == SCV_DIST_DEGENERATE_WRONG) v = (pp.truth + 500u) % 1000u;
                else v = (i < full) ? (base + 37u * (uint32_t)(i % m)) % 1000u : (base + 999u) % 1000u;
                arow[i] = (int32_t)v;
            }
            if (trow) trow[i] = (int32_t)(100u + mulhi32((uint32_t)(mix64(u ^ kGolden) >> 32), 11901u));
        }
    }
    // problems with B == 0 or no cells still need their truth
    if (truth && B == 0)
        for (int64_t pl = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; pl < P; pl += (int64_t)gridDim.x * blockDim.x)
            truth[pl] = (int32_t)problem_params(seed, p_offset + pl).truth;
}

// The device error word as an int64 in caller memory, in stream order (scv_export_error_word): lets a multi-rank
// caller put it behind the counters of the SAME all-reduce without a host round trip.
__global__ void scv_export_err_k(const uint32_t* err_flag, uint32_t mask, long long* dst) { *dst = (long long)(*err_flag & mask); }

// ---- problem-level bootstrap --------------------------------------------------------------------

__global__ __launch_bounds__(256) void scv_bootstrap_k(const scv_cell* cells, int64_t P, int32_t B, int32_t r_begin,
                                                       uint64_t seed, int32_t M, unsigned long long* out,
                                                       uint32_t* err_flag) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];
    const int32_t r = r_begin + (int32_t)blockIdx.x;
    for (int i = threadIdx.x; i < B * M; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    bool overflow = false;
    const uint4* c4 = reinterpret_cast<const uint4*>(cells);
    for (int64_t j = threadIdx.x; j < P; j += blockDim.x) {
        const uint64_t u = mix64(seed + kGolden * ((uint64_t)r * (uint64_t)P + (uint64_t)j + 1));
        const int64_t idx = (int64_t)mulhi32((uint32_t)(u >> 32), (uint32_t)P);
        for (int32_t b = 0; b < B; ++b) {
            const uint4 c = c4[idx * B + b];
            if (c.w & 0xffu) {
                const uint32_t nm = c.z & 0xffffu;
                if (nm >= (uint32_t)M) overflow = true;
                else atomicAdd(&cnt[b * M + nm], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * M; i += blockDim.x)
        out[(int64_t)blockIdx.x * B * M + i] = cnt[i];
    if (overflow) atomicOr(err_flag, 2u);
}

// LDS-resident variant: all the bootstrap looks at is (hit, n_modes) per cell -- 2 bytes.  Every
// workgroup stages the whole [P, B] table once as u16 codes (hit ? n_modes : 0; 20 KB at P = 10^4, B = 1) and then
// runs several resamples out of LDS: no 16-byte gathers from global per draw.  Class-1 hits (a strict win, by
// far the commonest code) are counted per wave with a ballot + popcount and ONE LDS atomic per wave, so the 64
// lanes of a wave do not serialise on the same counter; the rare tie classes use per-lane LDS atomics.
// Same draws and the same int64 [R, B, M] output as scv_bootstrap_k / oracle scvo_bootstrap.
__global__ __launch_bounds__(1024) void scv_bootstrap_lds_k(const scv_cell* cells, int64_t P, int32_t B, int32_t r_begin,
                                                            int32_t r_end, uint64_t seed, int32_t M, unsigned long long* out,
                                                            uint32_t* err_flag) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* cnt = lds;                                                   // [B * M]
    uint16_t* tab = reinterpret_cast<uint16_t*>(lds + (((int64_t)B * M + 3) & ~(int64_t)3));   // [P * B]
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
    const int64_t ncells = P * B;
    const uint4* c4 = reinterpret_cast<const uint4*>(cells);
    for (int64_t i = tid; i < ncells; i += T) {
        const uint4 c = c4[i];
        tab[i] = (c.w & 0xffu) ? (uint16_t)(c.z & 0xffffu) : (uint16_t)0;
    }
    bool overflow = false;
    const int64_t BM = (int64_t)B * M;
    for (int32_t r = r_begin + (int32_t)blockIdx.x; r < r_end; r += (int32_t)gridDim.x) {
        for (int64_t i = tid; i < BM; i += T) cnt[i] = 0;
        __syncthreads();                                                   // table staged (first pass) / counters zero
        // uniform trip count per wave (the ballot needs the whole wave): draws past P contribute nothing.
        // The generator argument seed + G * (r * P + j + 1) advances by G * T per iteration: one 64-bit add
        // instead of a 64-bit multiply (integer multiplies are quarter rate and this kernel is bound by them).
        uint64_t arg = seed + kGolden * ((uint64_t)r * (uint64_t)P + (uint64_t)tid + 1);
        const uint64_t darg = kGolden * (uint64_t)T;
        for (int64_t j0 = (int64_t)(tid - lane); j0 < P; j0 += T, arg += darg) {
            const int64_t j = j0 + lane;
            const bool live = j < P;
            const uint64_t u = mix64(arg);
            const int64_t idx = live ? (int64_t)mulhi32((uint32_t)(u >> 32), (uint32_t)P) : 0;
            for (int32_t b = 0; b < B; ++b) {
                const uint32_t code = live ? (uint32_t)tab[idx * B + b] : 0u;
                const uint64_t ones = __ballot(code == 1u);
                if (lane == 0 && ones) {
                    if (M > 1) atomicAdd(&cnt[(int64_t)b * M + 1], (uint32_t)__popcll(ones));
                }
                if (code == 1u && M <= 1) overflow = true;
                if (code > 1u) {
                    if (code >= (uint32_t)M) overflow = true;
                    else atomicAdd(&cnt[(int64_t)b * M + code], 1u);
                }
            }
        }
        __syncthreads();
        unsigned long long* o = out + (int64_t)(r - r_begin) * BM;
        for (int64_t i = tid; i < BM; i += T) o[i] = cnt[i];
        __syncthreads();                                                   // before the next resample re-zeroes cnt
    }
    if (overflow) atomicOr(err_flag, 2u);
}
#endif  // SCV_TU_MAIN

}  // namespace scv
