// scvote_prefix.hip.h -- prefix budgets over ONE sample pool per problem, every budget out of ONE pass (scv_prefix_pool).
//
// The reference's own data shape (/root/reference/o1.py:274-277 with the idx-keyed cache of o1.py:85-88): the budgets
// T = 2^11, 2^12, 2^13 ... of a problem are majority votes over the first N_b = 1, 2, 4 ... samples of the SAME list of
// completions.  Dense [P, B, N] (or the cell kernels on pool rows, one cell per (problem, budget)) re-read and re-count the
// prefix B times; here a problem's pool row is read ONCE and every vote is counted ONCE, whatever the number of budgets.
//
// How (pools of up to 4096 votes; a problem occupies G = 16 / 32 adjacent lanes, 64 / G problems per wave):
//  * the problem has a private 1024-bin histogram in LDS (32-bit bins at G = 32; 16-bit bins, two per word, at G = 16: 8 KiB of
//    histograms per wave instead of 16, so 16 waves per CU are resident instead of 8).  Vote i goes in with ONE returning LDS atomic
//    (ds_add_rtn_u32): the returned old count + 1 is the vote's RANK r_i = #{ j <= i in serialisation order : x_j == x_i }.
//    Among the votes of a prefix every bin with final count c contributes the ranks 1 .. c exactly once, so with
//    M = max rank over the prefix:
//        max_count = M        len(statistics.multimode) = #{ votes of the prefix with rank == M }
//        min(modes) = min{ x_i : r_i == M }
//    -- order-free facts about the multiset of (value, rank) pairs: the lanes of a problem add their votes concurrently, the
//    only ordering that matters is that the votes below a budget's boundary are all in before its snapshot and none above.
//  * every lane keeps two running registers over ITS votes: K = max (r << 18 | A), A = the LDS byte address of the vote's bin in a
//    histogram stored by DESCENDING value (the group maximum is M and the smallest modal value: the address doubles as the value,
//    12 VALU per vote at G = 32, 15 with the half-word arithmetic of G = 16), and S = (its largest rank << 18 | how many of its
//    votes have that rank).  A boundary costs one
//    group all-reduce of K, one of S's count where the lane's rank equals M (packed with the lane's truth votes), and for
//    the tokens stream one 64-bit sum -- not a re-read and a re-count.  The reduced pair is LATCHED by one lane of the
//    problem; after G boundaries (or the last) the lanes turn their pairs into records side by side, write them --
//    cells[p, b] for consecutive b are 16 bytes apart -- and add to the per-workgroup LDS counter tables (o1.py:238-240 as
//    integers), flushed once per workgroup.
//  * THE HEAD: budgets of up to 16 votes (the reference's 1, 2, 4, 8, 16) come out of ONE inclusive scan, not one reduction
//    each.  Lane l < 16 of the problem loads vote l; its rank IN INDEX ORDER is 1 + #{ j < l : x_j == x_l } (15 row_shr
//    compares inside the 16-lane DPP row), and the prefix statistics are a scan of (rank, 1) under
//        (M1, n1) + (M2, n2) = M1 > M2 ? (M1, n1) : M2 > M1 ? (M2, n2) : (M1, n1 + n2)
//    (associative, commutative; 4 row_shr steps) next to a running maximum of the keys and a running sum of the truth votes:
//    afterwards lane l holds the complete record of the prefix of l + 1 votes and the lane n_valid[b] - 1 writes budget b's.
//    The head's votes enter the histogram with one plain ds_add; the vectors below skip them.
//  * votes equal to the problem's TRUTH do not enter the histogram: the lane counts them (truth_count is needed anyway --
//    pass@k's c and the hit test of o1.py:206) and the boundary merges the two: max_count = max(M, truth votes), the truth
//    joins the modes on a tie.  The truth is the one value known before looking at the data that is usually the hot one;
//    up to G lanes adding to one bin would be serialised.  Inactive vote slots and truth votes add to a word of the lane's own.
//  * a row is held V 16-byte vectors per lane at a time (a CHUNK of 16 G V votes: the whole row up to 256 votes at G = 16, 512 at
//    G = 32), vector k of lane l = votes (k G + l) 4 .. + 3.  The NEXT chunk -- of this row, or the first of the next problem's
//    row with its truth and head -- is requested at the top of the current one into a second register set, so every load has a
//    whole chunk's work to land.  The record stores of a problem's LAST boundaries are issued at the top of the next problem, right
//    behind those requests: hipcc cannot count the vector-memory operations of the dynamic boundary loops and waits for the
//    next chunk with s_waitcnt vmcnt(0), which then finds nothing younger than a chunk's work (a store issued at the END of a
//    problem would be waited for whole at the top of the next).  Loads are unconditional (a vector beyond the longest budget
//    re-reads vector 0: a cache hit).
//  * boundaries are visited in ascending order (rank sort of n_valid in LDS: unsorted / duplicate / empty budgets are fine) and are
//    wave-uniform (n_valid is per budget, not per problem), so all control flow below is scalar: a vector is counted whole (no
//    masks) unless a boundary cuts it.
//
// Algorithmic bytes: 4 per vote of the pool (8 with tokens), 16 written per (problem, budget) -- DESIGN.md 3.8.
#pragma once

#include "scvote_kernels.hip.h"
#include "scvote_dispatch.h"

namespace scv {

#ifndef SCV_PREFIX_H16
#define SCV_PREFIX_H16 1                            // 0: 32-bit bins at G = 16 too (A/B builds)
#endif
// waves per workgroup = what the LDS of a CU (64 / G histograms per wave) and the registers (the token variants hold two more register sets) allow
template <int G, bool TOK>
constexpr int prefix_pool_waves() { return TOK ? 8 : (G == 16 ? (prefix_pool_h16(16) ? 16 : 8) : 12); }
constexpr int kRankShift = 18;                      // keys: rank << 18 | LDS byte address (< 160 KiB); ranks are <= 4096

constexpr int kPrefixHead = 16;                     // votes served by the head scan (one DPP row)

__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t addr) {
    return __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>((uintptr_t)addr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int SH>
__device__ __forceinline__ uint32_t row_shr_or(uint32_t v, uint32_t fill) {     // lane l <- lane l - SH of its 16-lane row; `fill` where there is none
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x110 + SH, 0xf, 0xf, false);
}
// 1 + number of lower lanes of the row holding the same value (values that must match nothing are made unique by the caller)
__device__ __forceinline__ uint32_t row_rank_in_order(uint32_t xs) {
    uint32_t r = 1u;
#define SCV_RANK_STEP(k) r += (row_shr_or<k>(xs, 0xfffffffeu) == xs) ? 1u : 0u;
    SCV_RANK_STEP(1) SCV_RANK_STEP(2) SCV_RANK_STEP(3) SCV_RANK_STEP(4) SCV_RANK_STEP(5) SCV_RANK_STEP(6) SCV_RANK_STEP(7) SCV_RANK_STEP(8)
    SCV_RANK_STEP(9) SCV_RANK_STEP(10) SCV_RANK_STEP(11) SCV_RANK_STEP(12) SCV_RANK_STEP(13) SCV_RANK_STEP(14) SCV_RANK_STEP(15)
#undef SCV_RANK_STEP
    return r;
}
// (max rank << 18 | votes at that rank) of two disjoint sets of votes
__device__ __forceinline__ uint32_t rank_count_merge(uint32_t a, uint32_t b) {
    const uint32_t hi = a > b ? a : b;
    return ((a ^ b) < (1u << 18)) ? a + (b & ((1u << 18) - 1u)) : hi;
}
template <int SH>
__device__ __forceinline__ void head_scan_step(uint32_t& KS, uint32_t& WS, uint32_t& TS) {
    const uint32_t k = row_shr_or<SH>(KS, 0u);
    KS = k > KS ? k : KS;
    WS = rank_count_merge(WS, row_shr_or<SH>(WS, 0u));
    TS += row_shr_or<SH>(TS, 0u);
}
// inclusive prefix sum over the row of a 64-bit value, as three limbs of 22 / 21 / 21 bits (16 x 2^22 < 2^32: no carries across lanes)
__device__ __forceinline__ long long row_prefix_sum_i64(long long v) {
    const unsigned long long u = (unsigned long long)v;
    uint32_t s0 = (uint32_t)(u & 0x3fffffu), s1 = (uint32_t)((u >> 22) & 0x1fffffu), s2 = (uint32_t)((u >> 43) & 0x1fffffu);
#define SCV_SUM_STEP(k) s0 += row_shr_or<k>(s0, 0u); s1 += row_shr_or<k>(s1, 0u); s2 += row_shr_or<k>(s2, 0u);
    SCV_SUM_STEP(1) SCV_SUM_STEP(2) SCV_SUM_STEP(4) SCV_SUM_STEP(8)
#undef SCV_SUM_STEP
    return (long long)((unsigned long long)s0 + ((unsigned long long)s1 << 22) + ((unsigned long long)s2 << 43));
}

// compiler barrier that also pins the registers of `r` (their values exist before the statement, memory operations stay behind it)
template <typename T, int NREG>
__device__ __forceinline__ void pin_before_loads(T (&r)[NREG]) {
    if constexpr (NREG >= 8) {
#pragma unroll
        for (int i = 0; i + 8 <= NREG; i += 8)
            asm volatile("" : "+v"(r[i]), "+v"(r[i + 1]), "+v"(r[i + 2]), "+v"(r[i + 3]), "+v"(r[i + 4]), "+v"(r[i + 5]), "+v"(r[i + 6]), "+v"(r[i + 7]) : : "memory");
    } else asm volatile("" : : : "memory");
}

// G lanes per problem, V vectors per lane per chunk; TOK: tokens stream; VEC: every pool row is 16-byte aligned (N % 4 == 0, aligned bases).
// LDS: per wave 64 / G x 1024 bins + kPrefixPoolLaneWords; behind the waves: ord[B] | nvs[B] | head map (32 words) | counter tables.
template <int G, int V, bool TOK, bool VEC>
__global__ __launch_bounds__((64 * prefix_pool_waves<G, TOK>())) void scv_prefix_pool(const AggArgs a) {
    constexpr int C = 64 / G;                  // problems per wave
    constexpr int VB = 4 * G;                  // votes of a problem per vector
    constexpr int CH = VB * V;                 // votes of a problem per chunk
    constexpr bool H16 = prefix_pool_h16(G);   // 16-bit bins, two per word
    constexpr int BS = H16 ? 1 : 2;            // log2(bytes per bin)
    constexpr int HW = prefix_pool_hist_words(G);                    // histogram words per wave
    constexpr uint32_t RK1 = 1u << kRankShift;
    constexpr int HEAD = kPrefixHead;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_wg[];
    const int tid = (int)threadIdx.x, T = (int)blockDim.x;
    const int lane = tid & 63, nw = T >> 6;
    const int sub = lane / G, l = lane % G;
    uint32_t* smem = smem_wg + (tid >> 6) * a.wave_lds_words;
    int32_t* ord = reinterpret_cast<int32_t*>(smem_wg + (int64_t)nw * a.wave_lds_words);   // budgets by ascending n_valid
    int32_t* nvs = ord + a.B;                                                             // their n_valid, ascending
    int32_t* hmap = nvs + a.B;                 // [0, 16): the budget whose n_valid is i + 1 (-1: none); [16, 32): how many budgets have it
    if (sort_prefix_took_it(a, tid, T)) return;
    {
        uint4* h4 = reinterpret_cast<uint4*>(smem);
        for (int i = lane; i < (HW + kPrefixPoolLaneWords) / 4; i += 64) h4[i] = make_uint4(0, 0, 0, 0);
    }
    if (tid < 32) hmap[tid] = tid < HEAD ? -1 : 0;
    __syncthreads();
    for (int b0 = 0; b0 < a.B; b0 += T) {
        const int b = b0 + tid;
        const bool have = b < a.B;
        const int64_t nb = have ? valid_len(a, b) : 0;
        const int rank = budget_rank<false>(a, b, nb);
        if (have) {
            ord[rank] = b;
            nvs[rank] = (int32_t)nb;
            if (nb >= 1 && nb <= HEAD) {
                hmap[nb - 1] = b;
                atomicAdd(reinterpret_cast<uint32_t*>(hmap) + HEAD + (nb - 1), 1u);
            }
        }
    }
    const WgCounters wgc = wg_counters_begin(a, reinterpret_cast<uint32_t*>(hmap + kPrefixPoolFixedWords), tid, T);
    __syncthreads();

    const uint32_t base = (uint32_t)(uintptr_t)(lds_u32*)smem;       // LDS byte offset of this wave's region
    const uint32_t hbase = base + (uint32_t)sub * ((uint32_t)kBins << BS);              // this problem's histogram
    const uint32_t htop = hbase + (1023u << BS);                     // the bin of value 0: value x lives at htop - (x << BS)
    const uint32_t trash = base + (uint32_t)HW * 4u + (uint32_t)lane * 4u;
    const int32_t B = a.B;
    const int32_t nmax = __builtin_amdgcn_readfirstlane(B > 0 ? nvs[B - 1] : 0);          // votes of the longest budget
    // budgets without votes (kz of them, first in the order), budgets the head serves (up to kh); is some head length asked for twice?
    int32_t kz = 0, kh = 0;
    bool head_dup = false;
    {
        uint32_t z = 0, h = 0;
        for (int k = lane; k < B; k += 64) { const int32_t nv = nvs[k]; z += nv == 0; h += nv <= HEAD; }
        kz = (int32_t)wave_sum_u32(z);
        kh = (int32_t)wave_sum_u32(h);
        head_dup = __any(lane < HEAD && hmap[HEAD + (lane & (HEAD - 1))] > 1);
    }
    const int32_t my_head_budget = (l < HEAD) ? hmap[l] : -1;        // the budget this lane's head prefix answers (no duplicates)
    const int32_t nchunks = (nmax + CH - 1) / CH;
    const int64_t nwaves = (int64_t)gridDim.x * nw;
    const int64_t wave = (int64_t)blockIdx.x * nw + (tid >> 6);
    const int64_t nbatches = (a.P + C - 1) / C;

    uint32_t cur[4 * V], nxt[4 * V];           // the chunk being counted / the chunk in flight: vector k in [4 k, 4 k + 3]
    int32_t tcur[TOK ? 4 * V : 1], tnxt[TOK ? 4 * V : 1];            // their tokens
    uint32_t hv_n = 0;                         // in flight: the next problem's head (vote l of the row, lanes l < 16), its token, its truth
    int32_t ht_n = 0, truth_n = 0;
    // chunk c of a row -> nxt: loads only
    auto load_chunk = [&](const int32_t* row, const int32_t* trow, int32_t c) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int32_t vi = (c * V + k) * G + l;                  // in 16-byte units
            if (VEC) {
                const int32_t vj = vi * 4 < nmax ? vi : 0;
                const int4 x = stream_load(reinterpret_cast<const int4*>(row) + vj);
                nxt[4 * k] = (uint32_t)x.x; nxt[4 * k + 1] = (uint32_t)x.y; nxt[4 * k + 2] = (uint32_t)x.z; nxt[4 * k + 3] = (uint32_t)x.w;
                if constexpr (TOK) {
                    const int4 y = stream_load(reinterpret_cast<const int4*>(trow) + vj);
                    tnxt[4 * k] = y.x; tnxt[4 * k + 1] = y.y; tnxt[4 * k + 2] = y.z; tnxt[4 * k + 3] = y.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int32_t idx = vi * 4 + q;
                    const int32_t ii = idx < nmax ? idx : 0;
                    nxt[4 * k + q] = (uint32_t)__builtin_nontemporal_load(row + ii);
                    if constexpr (TOK) tnxt[4 * k + q] = __builtin_nontemporal_load(trow + ii);
                }
            }
        }
    };
    auto row_of = [&](int64_t bt) -> int64_t {
        const int64_t p = bt * C + sub;
        return p < a.P ? p : 0;
    };
    auto load_head = [&](int64_t r) {
        const int32_t ii = (l < HEAD && l < nmax) ? l : 0;
        hv_n = (uint32_t)__builtin_nontemporal_load(a.answers + r * a.N + ii);
        if constexpr (TOK) ht_n = __builtin_nontemporal_load(a.tokens + r * a.N + ii);
        truth_n = a.truth[r];
    };

    // o1.py:204-213 + statistics.py:599-601 from a reduced pair; the record and the counters of (problem pp, budget b) by ONE lane
    auto emit = [&](int64_t pp, uint32_t tcmp, int32_t b, uint32_t gK, uint32_t packed, uint32_t pc, uint32_t pval, long long tok) {
        const uint32_t Mh = gK >> kRankShift;                         // largest count among the values that are neither the truth nor the pivot
        const uint32_t tc = packed & 0xffffu;                         // votes for the truth (<= 4096)
        const uint32_t nmh = Mh ? packed >> 16 : 0u;                  // values (of the histogram) whose count is Mh (<= 1023)
        uint32_t M = Mh > tc ? Mh : tc;
        M = pc > M ? pc : M;                                          // pc: votes for the problem's pivot value pval
        const uint32_t hit = (tc == M && M > 0u) ? 1u : 0u;           // o1.py:206 (multimode([]) == []: no hit)
        const uint32_t pin = (pc == M && pc > 0u) ? 1u : 0u;
        const uint32_t nm = (Mh == M ? nmh : 0u) + hit + pin;
        uint32_t mm = Mh == M ? (htop - (gK & (RK1 - 1u))) >> BS : 1024u;
        if (hit && tcmp < mm) mm = tcmp;
        if (pin && pval < mm) mm = pval;
        const int64_t cell = pp * B + b;
        if (a.cells) reinterpret_cast<uint4*>(a.cells)[cell] = make_uint4(M, tc, (nm & 0xffffu) | ((M ? (mm & 0xffffu) : 0xffffu) << 16), hit);
        if (TOK && a.cell_tokens) a.cell_tokens[cell] = tok;
        if (wgc.tcl > 0) wg_counters_add<TOK>(a, wgc, b, hit, nm, tc, tok);
        else {
            if (a.tie_hits && hit) atomicAdd(&a.tie_hits[(int64_t)b * SCV_TIE_CLASSES + nm], 1ull);
            if (TOK && a.token_sum) atomicAdd(&a.token_sum[b], (unsigned long long)tok);
            if (a.truth_sum) atomicAdd(&a.truth_sum[b], (unsigned long long)tc);
        }
    };
    // the pairs of the previous problem's last boundaries, latched and not yet written (lane j < Dn holds one)
    uint32_t DK = 0, DP = 0;
    uint32_t Dbp = 0;                          // budget | pivot votes << 16
    uint32_t Dtp = 0;                          // the problem's truth (0xffff: none) | its pivot value << 16 | pairs held << 26
    long long Dtok = 0;

    if (wave < nbatches && nmax > 0) {
        const int64_t r0 = row_of(wave);
        load_head(r0);
        load_chunk(a.answers + r0 * a.N, TOK ? a.tokens + r0 * a.N : nullptr, 0);
    }
    for (int64_t bt = wave; bt < nbatches; bt += nwaves) {
        const int64_t p = bt * C + sub;
        const bool live = p < a.P;
        const int64_t prow = live ? p : 0;
        const int32_t* row = a.answers + prow * a.N;
        const int32_t* trow = TOK ? a.tokens + prow * a.N : nullptr;
        const int64_t rnext = row_of(bt + nwaves < nbatches ? bt + nwaves : bt);          // (the last batch re-reads its own row)
        uint32_t tcmp = 0xffffffffu;

        // running state of this lane over its votes of the prefix so far
        uint32_t S = 0;                        // largest rank of a vote of this lane << 16 | number of its votes with that rank
        uint32_t K = 0;                        // max over its votes of rank << 10 | 1023 - value
        uint32_t tcl = 0;                      // its votes equal to the truth
        uint32_t pcl = 0;                      // its votes equal to the problem's pivot
        uint32_t pv = 0xffffffffu;             // the pivot: the most frequent value among the first 16 votes that are not the truth
        long long tsum = 0;                    // its tokens
        // one vote after its atomic has returned: Tr = its rank << 18 (0: not counted), A = its bin's address
        auto post = [&](uint32_t A, uint32_t Tr) {
            const uint32_t S1 = S > Tr ? S : Tr;
            S = S1 + (((S1 ^ Tr) < RK1) ? 1u : 0u);                  // (Tr == 0 while the lane's rank is 0: a count nobody reads)
            const uint32_t key = Tr | A;                             // Tr == 0: below every real key
            K = key > K ? key : K;
        };
        // the returning atomic of one vote (`use`: active and not the truth) -> its rank << 18, 0 when it does not count
        auto count_vote = [&](uint32_t A, bool use) -> uint32_t {
            uint32_t x;
            if (H16) {
                const uint32_t inc = __builtin_amdgcn_alignbyte(1u, 1u, A);                 // 1 or 1 << 16 by the half of the word
                const uint32_t old = __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>((uintptr_t)(use ? (A & ~3u) : trash)), inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                x = __builtin_amdgcn_alignbyte(0u, old, A);          // the bin's half in the low 16 bits (the shift below drops the rest: counts are < 2^14)
            } else x = lds_add_rtn(use ? A : trash);
            const uint32_t Tr = (x << kRankShift) + RK1;
            return use ? Tr : 0u;
        };
        // vector k (its first vote is vote `vb` of the row): the votes with lo <= index < hi (FULL: all of them, no masks)
        auto pass = [&](int k, int32_t vb, int32_t lo, int32_t hi, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            // domain check of the whole vector up front (o1.py:140 int(extracted_answer) is unbounded; the extractor maps it into bins 0..1023):
            // the clamp runs only in the -- wave-uniform, rare -- case that some slot, counted or not, holds a larger value
            uint32_t vq[4] = {cur[4 * k], cur[4 * k + 1], cur[4 * k + 2], cur[4 * k + 3]};
            if (__any((vq[0] | vq[1] | vq[2] | vq[3]) > 1023u)) {
                uint32_t bad = 0;                                    // (reported right here: an accumulator across the kernel is a register the vote path needs)
#pragma unroll
                for (int q = 0; q < 4; ++q) {                        // (cur[] keeps the value: a later pass over this vector may be the one that counts it)
                    const bool act = FULL || (uint32_t)(vb + 4 * l + q - lo) < (uint32_t)(hi - lo);
                    bad |= act ? vq[q] : 0u;
                    vq[q] = vq[q] < 1023u ? vq[q] : 1023u;
                }
                if (bad > 1023u) atomicOr(a.err_flag, 1u);
            }
            uint32_t A[4], Tr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t v = vq[q];
                const bool act = FULL || (uint32_t)(vb + 4 * l + q - lo) < (uint32_t)(hi - lo);
                const bool is_t = v == tcmp;
                A[q] = htop - (v << BS);
                const bool is_p = v == pv;
                tcl += (act && is_t) ? 1u : 0u;
                pcl += (act && is_p) ? 1u : 0u;                      // (2 VALU per vote for the pivot; a second vote path for waves without one
                                                                     //  did not fit the registers of 16 waves per CU)
                if constexpr (TOK) tsum += act ? (long long)tcur[4 * k + q] : 0ll;
                Tr[q] = count_vote(A[q], act && !is_t && !is_p);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) post(A[q], Tr[q]);
        };
        // the pair this lane has latched (the nl-th boundary since the last flush sits in lane nl)
        uint32_t LK = 0, LP = 0, Lbp = 0;      // Lbp: budget | pivot votes << 16 (budgets < 512, votes <= 4096)
        int32_t nl = 0;
        long long Ltok = 0;
        // boundary k: every vote below it is in, none above
        auto boundary = [&](int32_t k) {
            const uint32_t gK = cellgroup_max<G>(K);
            const uint32_t cnt = ((S ^ gK) < RK1) ? (S & (RK1 - 1u)) : 0u;        // this lane's votes at the group's maximum rank
            const uint32_t packed = cellgroup_sum<G>((cnt << 16) | tcl);
            const uint32_t gpc = cellgroup_sum<G>(pcl);
            long long gtok = 0;
            if constexpr (TOK) gtok = cellgroup_sum_i64<G>(tsum);
            const int32_t b = __builtin_amdgcn_readfirstlane(ord[k]);
            if (l == nl) { LK = gK; LP = packed; Lbp = (uint32_t)b | (gpc << 16); Ltok = gtok; }
            nl += 1;
            if (nl == G) {
                if (live) emit(p, tcmp, (int32_t)(Lbp & 0xffffu), LK, LP, Lbp >> 16, pv, Ltok);
                nl = 0;
            }
        };
        // the previous problem's last records (their stores land while this problem is counted)
        auto flush_deferred = [&]() {
            if ((uint32_t)l < (Dtp >> 26)) emit(p - nwaves * C, (Dtp & 0xffffu) == 0xffffu ? 0xffffffffu : (Dtp & 0xffffu), (int32_t)(Dbp & 0xffffu), DK, DP, Dbp >> 16, (Dtp >> 16) & 0x3ffu, Dtok);
            Dtp = 0;
        };

        int32_t kb = 0;
        if (nchunks == 0) {
            flush_deferred();
            for (; kb < kz; ++kb) boundary(kb);                        // every budget is empty
        }
        int32_t pos = 0;                       // votes [0, pos) are in
        int32_t nvk = 0x7fffffff;              // n_valid of boundary kb
        for (int32_t c = 0; c < nchunks; ++c) {
            // ---- top of a chunk: it has landed; request the one after it ---------------------------------------------------------
#pragma unroll
            for (int i = 0; i < 4 * V; ++i) { cur[i] = nxt[i]; if constexpr (TOK) tcur[i] = tnxt[i]; }
            uint32_t hv = 0;
            int32_t ht = 0;
            if (c == 0) {
                hv = hv_n; ht = ht_n;
                tcmp = (truth_n >= 0 && truth_n < kBins) ? (uint32_t)truth_n : 0xffffffffu;
            }
            // everything that was in flight is consumed HERE, before the next requests: a use behind them would make hipcc wait for them too
            pin_before_loads(cur); pin_before_loads(tcur);
            asm volatile("" : "+v"(hv), "+v"(ht), "+v"(tcmp) : : "memory");
            if (c + 1 < nchunks) load_chunk(row, trow, c + 1);
            else {
                load_head(rnext);
                load_chunk(a.answers + rnext * a.N, TOK ? a.tokens + rnext * a.N : nullptr, 0);
            }
            if (c == 0) {
                flush_deferred();
                for (; kb < kz; ++kb) boundary(kb);                    // budgets without votes
                // ---- the head: budgets of up to 16 votes out of one scan ----------------------------------------------------
                const bool act = l < HEAD && l < nmax;
                if (act && hv > 1023u) atomicOr(a.err_flag, 1u);
                const uint32_t hc = hv < 1023u ? hv : 1023u;
                const bool is_t = hc == tcmp;
                const bool cand = act && !is_t;
                const uint32_t r = row_rank_in_order(cand ? hc : (0xffff0000u | (uint32_t)lane));
                const uint32_t A = htop - (hc << BS);
                // THE PIVOT: the most frequent value among these votes that is not the truth.  Its votes -- here and in the vectors below --
                // are counted by the lanes like the truth's and never enter the histogram: a wrong majority, an exact tie with the truth or
                // one wrong answer throughout would otherwise send up to G returning atomics per instruction to ONE bin (measured before:
                // 4096-vote pools 50 us on D0 .. D2, 106 / 91 / 177 us on D3 / D4 / D5; with the pivot 53-56 and 69 / 54 / 53).  Ranks of the
                // other values do not depend on it.
                const uint32_t pk = cellgroup_max<G>(cand ? ((r << kRankShift) | A) : 0u);
                // (a value with fewer than 3 of the 16 votes is not hot: no pivot, and a wave without pivots runs the plain vote path)
                pv = (pk >> kRankShift) >= 3u ? (htop - (pk & (RK1 - 1u))) >> BS : 0xffffffffu;
                const bool is_p = cand && hc == pv;
                const bool use = cand && !is_p;
                if (H16) lds_add(use ? (A & ~3u) : trash, __builtin_amdgcn_alignbyte(1u, 1u, A));
                else lds_add(use ? A : trash, 1u);
                K = use ? (r << kRankShift) | A : 0u;
                S = use ? (r << kRankShift) | 1u : 0u;
                tcl = (act && is_t) ? 1u : 0u;
                pcl = is_p ? 1u : 0u;
                if constexpr (TOK) tsum = act ? (long long)ht : 0ll;
                if (kh > kz) {
                    uint32_t KS = K, WS = S, TS = tcl | (pcl << 16);                 // (truth and pivot votes: two sums of at most 16 in one register)
                    head_scan_step<1>(KS, WS, TS); head_scan_step<2>(KS, WS, TS); head_scan_step<4>(KS, WS, TS); head_scan_step<8>(KS, WS, TS);
                    long long tks = 0;
                    if constexpr (TOK) tks = row_prefix_sum_i64(tsum);
                    const uint32_t packed = ((WS & (RK1 - 1u)) << 16) | (TS & 0xffffu);
                    if (!head_dup) {
                        if (live && my_head_budget >= 0) emit(p, tcmp, my_head_budget, KS, packed, TS >> 16, pv, tks);
                    } else {
                        for (int32_t k = kz; k < kh; ++k) {
                            const int32_t hi = __builtin_amdgcn_readfirstlane(nvs[k]);
                            const int32_t b = __builtin_amdgcn_readfirstlane(ord[k]);
                            if (live && l == hi - 1) emit(p, tcmp, b, KS, packed, TS >> 16, pv, tks);
                        }
                    }
                }
                kb = kh;
                pos = HEAD < nmax ? HEAD : nmax;
                nvk = kb < B ? __builtin_amdgcn_readfirstlane(nvs[kb]) : 0x7fffffff;
            }
            // ---- the chunk's vectors ----------------------------------------------------------------------------------------
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int32_t vb = c * CH + k * VB, ve = vb + VB;
                while (nvk <= ve) {                                  // boundaries inside (or at the end of) this vector
                    if (nvk > pos) { pass(k, vb, pos, nvk, std::false_type{}); pos = nvk; }
                    boundary(kb);
                    kb += 1;
                    nvk = kb < B ? __builtin_amdgcn_readfirstlane(nvs[kb]) : 0x7fffffff;
                }
                if (kb < B && pos < ve) {                            // a later boundary needs the rest of the vector
                    if (pos <= vb) pass(k, vb, vb, ve, std::true_type{});
                    else pass(k, vb, pos, ve, std::false_type{});
                    pos = ve;
                }
            }
        }
        // the boundaries latched since the last flush are written at the top of the next problem
        DK = LK; DP = LP; Dbp = Lbp; Dtok = Ltok;
        Dtp = (tcmp & 0xffffu) | ((pv & 0x3ffu) << 16) | ((live ? (uint32_t)nl : 0u) << 26);
        // the problem's histogram is zero again before the next batch's first vote (LDS operations of a wave execute in order)
        if (nmax > 0) {
            lds_v4u* h4 = reinterpret_cast<lds_v4u*>((uintptr_t)hbase);
#pragma unroll
            for (int i = 0; i < (kBins << BS) / 16 / G; ++i) h4[i * G + l] = scv_v4u{0u, 0u, 0u, 0u};
        }
        __builtin_amdgcn_wave_barrier();
    }
    if ((uint32_t)l < (Dtp >> 26)) {           // the last problem of this wave: batch wave + (its batches - 1) * nwaves
        const int64_t last_bt = wave + ((nbatches - 1 - wave) / nwaves) * nwaves;
        emit(last_bt * C + sub, (Dtp & 0xffffu) == 0xffffu ? 0xffffffffu : (Dtp & 0xffffu), (int32_t)(Dbp & 0xffffu), DK, DP, Dbp >> 16, (Dtp >> 16) & 0x3ffu, Dtok);
    }
    wg_counters_flush<TOK>(a, wgc, tid, T);
}

}  // namespace scv
