// scvote_prefix.hip.h -- prefix budgets over ONE sample pool per problem, every budget out of ONE pass (scv_prefix_pool).
//
// The reference's own data shape (/root/reference/o1.py:274-277 with the idx-keyed cache of o1.py:85-88): the budgets
// T = 2^11, 2^12, 2^13 ... of a problem are majority votes over the first N_b = 1, 2, 4 ... samples of the SAME list of
// completions.  Dense [P, B, N] (or the cell kernels on pool rows, one cell per (problem, budget)) re-read and re-count the
// prefix B times; here a problem's pool row is read ONCE and every vote is counted ONCE, whatever the number of budgets.
//
// How (pools of up to 4096 votes; a problem occupies G = 16 / 32 / 64 adjacent lanes, 64 / G problems per wave):
//  * the problem has a private 1024-bin histogram in LDS (32-bit bins).  Vote i goes in with ONE returning LDS atomic
//    (ds_add_rtn_u32): the returned old count + 1 is the vote's RANK r_i = #{ j <= i in serialisation order : x_j == x_i }.
//    Among the votes of a prefix every bin with final count c contributes the ranks 1 .. c exactly once, so with
//    M = max rank over the prefix:
//        max_count = M        len(statistics.multimode) = #{ votes of the prefix with rank == M }
//        min(modes) = min{ x_i : r_i == M }
//    -- order-free facts about the multiset of (value, rank) pairs: the lanes of a problem add their votes concurrently, the
//    only ordering that matters is that the votes below a budget's boundary are all in before its snapshot and none above.
//  * every lane keeps two running registers over ITS votes: K = max (r << 10 | 1023 - x) (the group maximum is M and the
//    smallest modal value) and S = (its largest rank << 16 | how many of its votes have that rank).  A boundary costs one
//    group all-reduce of K, one of S's count where the lane's rank equals M (packed with the lane's truth votes), and for
//    the tokens stream one 64-bit sum: ~16 wave instructions per budget, not a re-read and a re-count.
//  * votes equal to the problem's TRUTH do not enter the histogram: the lane counts them (truth_count is needed anyway --
//    pass@k's c and the hit test of o1.py:206) and the boundary merges the two: max_count = max(M, truth votes), the truth
//    joins the modes on a tie.  The truth is the one value known before looking at the data that is usually the hot one;
//    up to G lanes adding to one bin would be serialised.  Inactive vote slots and truth votes add to a word of the lane's own.
//  * boundaries are visited in ascending order (rank sort of n_valid in LDS: unsorted / duplicate / empty budgets are fine) and are
//    wave-uniform (n_valid is per budget, not per problem), so all control flow below is scalar.  The first 4 G votes of a
//    row are held TRANSPOSED (slot q of lane l = vote q G + l, four dword loads): the reference's budgets 1, 2, 4 ... 2^k
//    then fall on whole slots from G upwards (and below G need only slot 0); later blocks are one dwordx4 per lane.
//  * the record of boundary k is latched by lane k % G of the problem; after G boundaries (or the last) the lanes write
//    their records side by side -- cells[p, b] for consecutive b are 16 bytes apart: one contiguous store per problem -- and
//    add to the per-workgroup LDS counter tables (o1.py:238-240 as integers), flushed once per workgroup.
//
// Algorithmic bytes: 4 per vote of the pool (8 with tokens), 16 written per (problem, budget) -- DESIGN.md 3.8.
#pragma once

#include "scvote_kernels.hip.h"
#include "scvote_dispatch.h"

namespace scv {

template <int G>
constexpr int prefix_pool_waves() { return G == 16 ? 8 : 16; }       // histograms: 64 / G x 4 KiB per wave

__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t addr) {
    return __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>((uintptr_t)addr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// G lanes per problem; TOK: tokens stream; VEC: every pool row is 16-byte aligned (N % 4 == 0 and aligned bases).
template <int G, bool TOK, bool VEC>
__global__ __launch_bounds__((64 * prefix_pool_waves<G>())) void scv_prefix_pool(const AggArgs a) {
    constexpr int C = 64 / G;                  // problems per wave
    constexpr int BLK = 4 * G;                 // votes of a problem per block (4 per lane)
    constexpr int HW = C * kBins;              // histogram words per wave
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_wg[];
    const int tid = (int)threadIdx.x, T = (int)blockDim.x;
    const int lane = tid & 63, nw = T >> 6;
    const int sub = lane / G, l = lane % G;
    uint32_t* smem = smem_wg + (tid >> 6) * a.wave_lds_words;
    int32_t* ord = reinterpret_cast<int32_t*>(smem_wg + (int64_t)nw * a.wave_lds_words);   // budgets by ascending n_valid
    int32_t* nvs = ord + a.B;                                                             // their n_valid, ascending
    {
        uint4* h4 = reinterpret_cast<uint4*>(smem);
        for (int i = lane; i < (HW + kPrefixPoolLaneWords) / 4; i += 64) h4[i] = make_uint4(0, 0, 0, 0);
    }
    for (int b = tid; b < a.B; b += T) {
        const int64_t nb = valid_len(a, b);
        int rank = 0;
        for (int c = 0; c < a.B; ++c) {
            const int64_t nc = valid_len(a, c);
            rank += (nc < nb) || (nc == nb && c < b);
        }
        ord[rank] = b;
        nvs[rank] = (int32_t)nb;
    }
    const WgCounters wgc = wg_counters_begin(a, reinterpret_cast<uint32_t*>(nvs + a.B), tid, T);
    __syncthreads();

    const uint32_t base = (uint32_t)(uintptr_t)(lds_u32*)smem;       // LDS byte offset of this wave's region
    const uint32_t hbase = base + (uint32_t)sub * (kBins * 4u);      // this problem's histogram
    const uint32_t trash = base + (uint32_t)HW * 4u + (uint32_t)lane * 4u;
    const int32_t N = (int32_t)a.N, B = a.B;
    const int32_t nmax = __builtin_amdgcn_readfirstlane(B > 0 ? nvs[B - 1] : 0);          // votes of the longest budget
    const int64_t nwaves = (int64_t)gridDim.x * nw;
    const int64_t wave = (int64_t)blockIdx.x * nw + (tid >> 6);
    const int64_t nbatches = (a.P + C - 1) / C;
    uint32_t bad = 0;

    uint32_t cur[4] = {0, 0, 0, 0}, nxt[4] = {0, 0, 0, 0};     // votes of the current / the next block
    int32_t tcur[4] = {0, 0, 0, 0}, tnxt[4] = {0, 0, 0, 0};     // their tokens
    // block j of problem row `row`: loads only (unconditional: a slot past the row re-reads element 0 and is never active)
    auto load_block = [&](const int32_t* row, const int32_t* trow, int32_t j, uint32_t (&v)[4], int32_t (&t)[4]) {
        if (j == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int32_t idx = q * G + l;
                const int32_t ii = idx < N ? idx : 0;
                v[q] = (uint32_t)__builtin_nontemporal_load(row + ii);
                if (TOK) t[q] = __builtin_nontemporal_load(trow + ii);
            }
        } else if (VEC) {
            int32_t vi = j * G + l;
            vi = vi * 4 < N ? vi : 0;
            const int4 x = stream_load(reinterpret_cast<const int4*>(row) + vi);
            v[0] = (uint32_t)x.x; v[1] = (uint32_t)x.y; v[2] = (uint32_t)x.z; v[3] = (uint32_t)x.w;
            if (TOK) {
                const int4 y = stream_load(reinterpret_cast<const int4*>(trow) + vi);
                t[0] = y.x; t[1] = y.y; t[2] = y.z; t[3] = y.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int32_t idx = j * BLK + 4 * l + q;
                const int32_t ii = idx < N ? idx : 0;
                v[q] = (uint32_t)__builtin_nontemporal_load(row + ii);
                if (TOK) t[q] = __builtin_nontemporal_load(trow + ii);
            }
        }
    };
    auto row_of = [&](int64_t bt) -> int64_t {
        const int64_t p = bt * C + sub;
        return p < a.P ? p : 0;
    };

    if (wave < nbatches && nmax > 0) {
        const int64_t r0 = row_of(wave);
        load_block(a.answers + r0 * a.N, TOK ? a.tokens + r0 * a.N : nullptr, 0, cur, tcur);
        if (BLK < nmax) load_block(a.answers + r0 * a.N, TOK ? a.tokens + r0 * a.N : nullptr, 1, nxt, tnxt);
    }
    for (int64_t bt = wave; bt < nbatches; bt += nwaves) {
        const int64_t p = bt * C + sub;
        const bool live = p < a.P;
        const int64_t prow = live ? p : 0;
        const int32_t* row = a.answers + prow * a.N;
        const int32_t* trow = TOK ? a.tokens + prow * a.N : nullptr;
        const int32_t truth = a.truth[prow];
        const uint32_t tcmp = (truth >= 0 && truth < kBins) ? (uint32_t)truth : 0xffffffffu;

        // running state of this lane over its votes of the prefix so far
        uint32_t S = 0;                        // largest rank of a vote of this lane << 16 | number of its votes with that rank
        uint32_t K = 0;                        // max over its votes of rank << 10 | 1023 - value
        uint32_t tcl = 0;                      // its votes equal to the truth
        long long tsum = 0;                    // its tokens
        // one vote: `use` = counts for the histogram (active and not the truth)
        auto post = [&](uint32_t vc, uint32_t old, bool use) {
            const uint32_t r = use ? old + 1u : 0u;
            const uint32_t Tr = r << 16;
            const uint32_t S1 = S > Tr ? S : Tr;
            S = S1 + (((S1 ^ Tr) < 0x10000u) ? 1u : 0u);             // (r == 0 while the lane's rank is 0: a count nobody reads)
            const uint32_t key = (r << 10) | (1023u - vc);           // r == 0: below every real key
            K = key > K ? key : K;
        };
        // the four slots of block j, votes with lo <= index < hi (FULL: the whole block is inside, no masks)
        auto pass = [&](const uint32_t (&v)[4], const int32_t (&t)[4], int32_t j, int32_t lo, int32_t hi, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            uint32_t vc[4], old[4];
            bool use[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int32_t idx = j == 0 ? q * G + l : j * BLK + 4 * l + q;
                const bool act = FULL || (uint32_t)(idx - lo) < (uint32_t)(hi - lo);
                bad |= act ? v[q] : 0u;
                vc[q] = v[q] < 1023u ? v[q] : 1023u;
                const bool is_t = vc[q] == tcmp;
                use[q] = act && !is_t;
                tcl += (act && is_t) ? 1u : 0u;
                if (TOK) tsum += act ? (long long)t[q] : 0ll;
                old[q] = lds_add_rtn(use[q] ? hbase + (vc[q] << 2) : trash);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) post(vc[q], old[q], use[q]);
        };
        // one slot of a block (budgets below 4 G votes touch one slot at a time)
        auto pass_slot = [&](uint32_t v, int32_t t, int32_t idx, int32_t lo, int32_t hi) {
            const bool act = (uint32_t)(idx - lo) < (uint32_t)(hi - lo);
            bad |= act ? v : 0u;
            const uint32_t vc = v < 1023u ? v : 1023u;
            const bool is_t = vc == tcmp;
            const bool use = act && !is_t;
            tcl += (act && is_t) ? 1u : 0u;
            if (TOK) tsum += act ? (long long)t : 0ll;
            const uint32_t old = lds_add_rtn(use ? hbase + (vc << 2) : trash);
            post(vc, old, use);
        };

        // the record this lane has latched (boundary k of the current group of G boundaries sits in lane k % G)
        uint32_t rx = 0, ry = 0, rz = 0xffff0000u, rw = 0;
        long long rtok = 0;
        int32_t pos = 0;                       // votes [0, pos) are in
        int32_t cj = 0;                        // block held by `cur` (`nxt` holds cj + 1 when that block is needed)
        for (int32_t k = 0; k < B; ++k) {
            const int32_t hi = __builtin_amdgcn_readfirstlane(nvs[k]);
            while (pos < hi) {
                const int32_t blo = cj * BLK, bhi = blo + BLK;
                const int32_t seg = hi < bhi ? hi : bhi;
                if (pos == blo && seg == bhi) pass(cur, tcur, cj, blo, bhi, std::true_type{});
                else if (seg - pos >= 2 * G || cj > 0) pass(cur, tcur, cj, pos, seg, std::false_type{});
                else {
                    // block 0, a short range: only the slots it touches (slot q holds votes [q G, (q + 1) G))
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q * G < seg && (q + 1) * G > pos) pass_slot(cur[q], TOK ? tcur[q] : 0, q * G + l, pos, seg);
                }
                pos = seg;
                if (pos == bhi) {              // block consumed: the next one becomes current, the one after it is requested
#pragma unroll
                    for (int q = 0; q < 4; ++q) { cur[q] = nxt[q]; if (TOK) tcur[q] = tnxt[q]; }
                    cj += 1;
                    if ((cj + 1) * BLK < nmax) load_block(row, trow, cj + 1, nxt, tnxt);
                }
            }
            if (k == B - 1 && bt + nwaves < nbatches) {
                // every vote of this batch is in: the next batch's first blocks fly while the last boundary is reduced and the
                // histograms are cleared
                const int64_t rn = row_of(bt + nwaves);
                load_block(a.answers + rn * a.N, TOK ? a.tokens + rn * a.N : nullptr, 0, cur, tcur);
                if (BLK < nmax) load_block(a.answers + rn * a.N, TOK ? a.tokens + rn * a.N : nullptr, 1, nxt, tnxt);
            }
            // ---- boundary k: statistics.multimode of the prefix (statistics.py:599-601) + o1.py:204-213 ------------
            const uint32_t gK = cellgroup_max<G>(K);
            const uint32_t Mh = gK >> 10;                             // largest count among the values that are not the truth
            const uint32_t cnt = ((S >> 16) == Mh) ? (S & 0xffffu) : 0u;
            const uint32_t packed = cellgroup_sum<G>((cnt << 16) | tcl);
            const uint32_t tc = packed & 0xffffu;                     // votes for the truth (<= 4096)
            const uint32_t nmh = Mh ? packed >> 16 : 0u;              // values (not the truth) whose count is Mh (<= 1023)
            const uint32_t M = Mh > tc ? Mh : tc;
            const uint32_t hit = (tc == M && M > 0u) ? 1u : 0u;       // o1.py:206 (multimode([]) == []: no hit)
            const uint32_t nm = (Mh == M ? nmh : 0u) + hit;
            uint32_t mm = Mh == M ? 1023u - (gK & 1023u) : 1024u;
            if (hit && tcmp < mm) mm = tcmp;
            long long gtok = 0;
            if (TOK) gtok = cellgroup_sum_i64<G>(tsum);
            if (l == (k % G)) {
                rx = M; ry = tc; rw = hit;
                rz = (nm & 0xffffu) | ((M ? (mm & 0xffffu) : 0xffffu) << 16);
                rtok = gtok;
            }
            // ---- G boundaries latched (or the last one): the lanes write their records and count them ----------------
            if ((k % G) == G - 1 || k == B - 1) {
                const int32_t k0 = k - (k % G);
                if (live && k0 + l <= k) {
                    const int32_t b = ord[k0 + l];
                    const int64_t cell = p * B + b;
                    if (a.cells) reinterpret_cast<uint4*>(a.cells)[cell] = make_uint4(rx, ry, rz, rw);
                    if (TOK && a.cell_tokens) a.cell_tokens[cell] = rtok;
                    const uint32_t n_modes = rz & 0xffffu;
                    if (wgc.tcl > 0) wg_counters_add<TOK>(a, wgc, b, rw, n_modes, ry, rtok);
                    else {
                        if (a.tie_hits && rw) atomicAdd(&a.tie_hits[(int64_t)b * SCV_TIE_CLASSES + n_modes], 1ull);
                        if (TOK && a.token_sum) atomicAdd(&a.token_sum[b], (unsigned long long)rtok);
                        if (a.truth_sum) atomicAdd(&a.truth_sum[b], (unsigned long long)ry);
                    }
                }
            }
        }
        // the problem's histogram is zero again before the next batch's first vote (LDS operations of a wave execute in order)
        if (nmax > 0) {
            lds_v4u* h4 = reinterpret_cast<lds_v4u*>((uintptr_t)hbase);
#pragma unroll
            for (int i = 0; i < kBins / 4 / G; ++i) h4[i * G + l] = scv_v4u{0u, 0u, 0u, 0u};
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    wg_counters_flush<TOK>(a, wgc, tid, T);
}

}  // namespace scv
