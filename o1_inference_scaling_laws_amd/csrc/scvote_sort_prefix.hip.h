// scvote_sort_prefix.hip.h -- prefix budgets over SHORT pools (17 .. 128 votes: every pool the reference forms), the budgets of the reference itself: powers of two.
//
// The reference's sweep (/root/reference/o1.py:274-277): the budgets of a problem are majority votes over the first 1, 2, 4 ... N samples
// of ONE list of completions.  scv_sort_cells sorts a cell's votes in registers with Batcher's odd-even mergesort: after the merge phase
// p of that network every aligned block of 2 p wires is sorted -- in particular wires 0 .. 2 p - 1, the FIRST 2 p votes of the row.  A
// run scan over that block IS the majority vote of the budget 2 p: maj@2, 4, 8 ... NV / 2 fall out of the one sort that maj@N needs
// anyway.  One lane per problem, rows by LDS-DMA exactly as in scv_sort_cells (scvote_sort.hip.h): the pool row is
// read ONCE, sorted ONCE, and every budget costs a scan of its own block (~8 plain VALU per vote of the block) and one 16-byte record.
//
// Budget CLASSES of a launch (wave-uniform: n_valid is per budget, not per problem): 0 = no votes | 1 + j = the first 2^j votes,
// 2^j <= NV / 2 | the last = all N votes (NV / 2 < N <= NV: the host picks the shape).  A launch whose budgets are not all of this form
// leaves WITHOUT side effects (every workgroup finds the same verdict from n_valid): in DEVICE mode the host cannot read n_valid, so it
// queues this kernel AND the general one (scv_lane_prefix / scv_prefix_pool with a.skip_sortable = NV), which leaves when this one
// runs; in HOST mode it knows and queues one.  Budgets of a class share the record; their counters (o1.py:238-240 as integers) are kept
// per CLASS (registers + a [classes][NV + 1] table in LDS, whatever the number of budgets) and handed to the budgets at the end.
//
// Records of a step are packed (7 + 7 + 7 + 10 + 1 bits) and written at the top of the NEXT step, behind the wait for its image: a store
// issued at the end of a step would be waited for whole by that s_waitcnt vmcnt(0).
//
// Algorithmic bytes: 4 per vote of the pool (8 with tokens), 16 written per (problem, budget) (24 with tokens) -- DESIGN.md 3.8.
#pragma once

#include "scvote_sort.hip.h"

namespace scv {

// Token steps a wave WITHOUT a sort step in the sort's last, partial round takes before any other wave gets one (scv_sort_prefix2<true>): measured
// 0 .. 4 on 2e5 pools of 128 votes: 80.4 / 80.3 / 74.4 / 77.5 / 78.3 us, equal elsewhere (profiles/r06_prefix_token_steps_ab.log)
#ifndef SCV_RECORDS_LAST
#define SCV_RECORDS_LAST 1
#endif
#ifndef SCV_TOK_PASSES
#define SCV_TOK_PASSES 2
#endif

// exchanges of the merge phases p' <= pmax of sv_make_network<N>() (the generator's own loops)
template <int N>
constexpr int sv_phase_end(int pmax) {
    int n = 0;
    for (int p = 1; p < N && p <= pmax; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= N - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) ++n;
    return n;
}
template <int NP, int FROM, typename Tick, int... I>
__device__ __forceinline__ void sv_sort_halves_range(uint32_t (&R)[NP], Tick& tick, std::integer_sequence<int, I...>) {
    ((sv_ce(R[SvNet<NP>::net.a[FROM + I]], R[SvNet<NP>::net.b[FROM + I]]), tick(R[SvNet<NP>::net.a[FROM + I]])), ...);
}
// the lockstep network phase by phase; hook(integral_constant<2 p>) after phase p: wires 0 .. 2 p - 1 are sorted in both halves
template <int NP, int P, typename Tick, typename Hook>
__device__ __forceinline__ void sv_sort_phases(uint32_t (&R)[NP], Tick& tick, Hook& hook) {
    if constexpr (P < NP) {
        constexpr int from = sv_phase_end<NP>(P / 2), to = sv_phase_end<NP>(P);
        sv_sort_halves_range<NP, from>(R, tick, std::make_integer_sequence<int, to - from>{});
        hook(std::integral_constant<int, 2 * P>{});
        sv_sort_phases<NP, 2 * P>(R, tick, hook);
    }
}
// the halves of a lockstep-sorted register file into one ascending sequence (the tail of sv_sort for NP a power of two)
template <int NP, typename Tick>
__device__ __forceinline__ void sv_merge_halves(uint32_t (&R)[NP], Tick& tick) {
    static_assert(sv_pow2(NP), "power-of-two shapes");
#pragma unroll
    for (int r = 0; r < NP / 2; ++r) { sv_ce_cross(R[r], R[NP - 1 - r]); tick(R[r]); }
#pragma unroll
    for (int j = NP >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int l = r ^ j;
            if (l > r) { sv_ce(R[r], R[l]); tick(R[r]); }
        }
    }
}

struct BlockStats { uint32_t max_run, at_max, min_at_max; };
// statistics.multimode of the M sorted values in the LOW halves of R[0 .. M - 1]: key = (M - length of the run ending here) << 10 | value,
// the smallest key is the last element of the longest run with the smallest value; a maximal run reaches its length once
template <int M, int NP>
__device__ __forceinline__ BlockStats sv_scan_block(const uint32_t (&R)[NP]) {
    static_assert(M <= NP, "a block of the low halves");
    uint32_t key[M];
    uint32_t prev = 0xffffffffu, s = 0, km = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const uint32_t x = R[i] & 0xffffu;
        s = (x != prev) ? (uint32_t)i : s;
        key[i] = (((uint32_t)(M - 1 - i) + s) << 10) | x;
        km = key[i] < km ? key[i] : km;
        prev = x;
    }
    const uint32_t thr = km | 0x3ffu;
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) cnt += key[i] <= thr ? 1u : 0u;
    return BlockStats{(uint32_t)M - (km >> 10), cnt, km & 0x3ffu};
}

#ifdef SCV_SP_TIMELINE
// Measurement build only (tools/sort_prefix_timeline.py; never defined for the product library): shader cycles every wave spent in the
// phases of a step, summed over all waves: 0 wait for the copy | 1 rows -> registers, packed, truth prefix | 2 the previous records |
// 3 sort (+ the next copy's pieces) | 4 the block scans between the phases | 5 final scan | 6 loop control, prologue; [7] = steps
__device__ unsigned long long scv_sp_timeline[8];
#define SP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_now = __builtin_readcyclecounter(); \
                         __builtin_amdgcn_sched_barrier(0); tl[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define SP_STAMP(i) do { } while (0)
#endif
#ifdef SCV_SP_WALL
// the same launch on the 100 MHz wall clock: [0] = the first wave's start (min), [i] = the LAST wave to pass mark i (max): 1 a wave starts | 2 classes set up |
// 3 the first image has landed | 4 rows in registers | 5 sorted | 6 final scan | 7 all steps done | 8 last records out | 9 counters handed out
// (one row of 16 marks per wave, plain stores: atomics on one word serialise the launch)
constexpr int kSpWallWaves = 4096;
__device__ unsigned long long scv_sp_wall[kSpWallWaves * 16];
#define SP_WALL(i) do { __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0) scv_sp_wall[sp_wall_row * 16 + (i)] = (unsigned long long)wall_clock64(); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
#define SP_WALL_START() const int sp_wall_row = (int)(((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kSpWallWaves); SP_WALL(0); SP_WALL(1); \
    if ((threadIdx.x & 63) == 0) { uint32_t hw_id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id)); scv_sp_wall[sp_wall_row * 16 + 10] = 0x100000000ull | hw_id; }
#else
#define SP_WALL(i) do { } while (0)
#define SP_WALL_START() do { } while (0)
#endif

// ---- shared by the two kernels ---------------------------------------------------------------------------------------------------------
// The budget CLASSES of the launch: cb[c] = first position of class c in ordl (budgets by class, in LDS), cb[NC] = B; cbeg = the same in LDS for
// the hand-out at the end.  Returns false when some budget has no class: the launch is not this kernel's (every workgroup finds the same
// verdict; error bit 8 when the caller had promised such budgets).  Up to 64 budgets (every list of the reference): ONE coalesced load of
// n_valid per wave, then everything in registers -- counts by ballot, positions by mbcnt, one barrier; longer lists: ranks by the whole
// workgroup, counts by LDS atomics.  `zeroed` words of LDS behind ordl (the class tables) are cleared on the way.
template <int NV, int NC>
__device__ __forceinline__ bool sort_prefix_setup_classes(const AggArgs& a, int32_t* cbeg, int32_t* ordl, uint32_t* zero_from, int zero_words,
                                                          int32_t (&cb)[NC + 1]) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, T = (int)blockDim.x;
    const int32_t N = (int32_t)a.N, B = a.B;
    if (tid < 16) cbeg[tid] = 0;
    for (int i = tid; i < zero_words; i += T) zero_from[i] = 0;
    if (B <= 64) {
        const bool have = lane < B;
        const int cls = have ? sort_prefix_class<NV>(valid_len(a, lane), N) : NC;
        if (__any(have && cls < 0)) {                                 // (the general kernel queued behind this one takes the launch ...
            if (a.budgets_promised && tid == 0) atomicOr(a.err_flag, 8u);   //  ... unless the budgets were promised to be of this form)
            return false;
        }
        uint32_t pos = 0;
        cb[0] = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned long long m = __ballot(cls == c);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            pos = cls == c ? (uint32_t)cb[c] + before : pos;
            cb[c + 1] = cb[c] + (int32_t)__builtin_popcountll(m);
        }
        if (wid == 0 && have) ordl[pos] = lane;
        if (tid == 0) {
#pragma unroll
            for (int c = 0; c <= NC; ++c) cbeg[c] = cb[c];
        }
        __syncthreads();
        return true;
    }
    __syncthreads();
    int bad = 0;
    for (int b0 = 0; b0 < B; b0 += T) {
        const int b = b0 + tid;
        const bool have = b < B;
        const int c = have ? sort_prefix_class<NV>(valid_len(a, b), N) : 0;
        bad |= c < 0 ? 1 : 0;
        // (a list this kernel does not serve: the ranks are not used)
        const int rank = budget_rank_of<false>(a, b, (int64_t)c, [&](int o) { return sort_prefix_class<NV>(valid_len(a, o), N); });
        if (have && c >= 0 && rank >= 0 && rank < B) {
            ordl[rank] = b;
            atomicAdd(reinterpret_cast<uint32_t*>(cbeg) + c + 1, 1u);
        }
    }
    if (__syncthreads_or(bad)) {
        if (a.budgets_promised && tid == 0) atomicOr(a.err_flag, 8u);
        return false;
    }
    if (tid == 0) {
        for (int c = 1; c <= NC; ++c) cbeg[c] += cbeg[c - 1];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c <= NC; ++c) cb[c] = __builtin_amdgcn_readfirstlane(cbeg[c]);
    return true;
}
// The classes' counters (tie[NC][TC] hits by tie class, acc[NC] truth sums | acc[NC ..] token sums, all in LDS and complete: call behind a
// barrier) handed to their budgets: o1.py:238-240 as integers, one device atomic per non-zero word and budget.
template <int NC, int TC, bool TOK>
__device__ __forceinline__ void sort_prefix_hand_out_counters(const AggArgs& a, const int32_t* cbeg, const int32_t* ordl, const uint32_t* tie,
                                                              const unsigned long long* acc) {
    const int tid = threadIdx.x, T = (int)blockDim.x;
    const int32_t B = a.B;
    for (int64_t i = tid; i < (int64_t)B * TC; i += T) {
        const int32_t j = (int32_t)(i / TC), k = (int32_t)(i - (int64_t)j * TC);
        int c = 0;
        while (c < NC - 1 && j >= cbeg[c + 1]) ++c;
        const uint32_t v = tie[c * TC + k];
        if (v && a.tie_hits) atomicAdd(&a.tie_hits[(int64_t)ordl[j] * SCV_TIE_CLASSES + k], (unsigned long long)v);
    }
    for (int j = tid; j < B; j += T) {
        int c = 0;
        while (c < NC - 1 && j >= cbeg[c + 1]) ++c;
        if (a.truth_sum && acc[c]) atomicAdd(&a.truth_sum[ordl[j]], acc[c]);
        if (TOK && a.token_sum && acc[NC + c]) atomicAdd(&a.token_sum[ordl[j]], acc[NC + c]);
    }
}

// NV: capacity of the shape (32 / 64); host contract: NV / 2 < N <= NV, N % 4 == 0, 16-byte aligned bases, B <= kMaxSortedB,
// a.wave_lds_words = 64 * PS * 4 + 64, PS = (N / 4) | 1 (one image: with tokens it holds a step's votes, then its tokens).  LDS behind the waves' regions: class offsets [16] |
// budgets by class [B rounded to 4] | tie classes [classes][NV + 1] | truth sums [classes] | token sums [classes] (64-bit).
constexpr int sort_prefix_threads(int nv) { return 512; }
template <int NV, bool TOK>
__global__ __launch_bounds__(sort_prefix_threads(NV)) void scv_sort_prefix(const AggArgs a) {
    constexpr int NP = NV / 2, RSM = NV / 4;
    constexpr int QMAX = RSM + 1;
    constexpr int TC = NV + 1;
    constexpr int LG = sv_log2(NP);
    constexpr int NC = LG + 3, CF = LG + 2;                          // classes; the class of all N votes
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, T = (int)blockDim.x, NW = T >> 6;
    const int32_t N = (int32_t)a.N, B = a.B;
    const uint32_t RS = (uint32_t)N >> 2, PS = RS | 1u;
    const uint32_t rowbytes = (uint32_t)N * 4u;
    int32_t* cbeg = reinterpret_cast<int32_t*>(lds + (int64_t)NW * a.wave_lds_words);   // [c]: first position of class c in ordl; [NC] = B
    int32_t* ordl = cbeg + 16;                                                           // budgets by class
    uint32_t* tie = reinterpret_cast<uint32_t*>(ordl + ((B + 3) & ~3));                  // [NC][TC]
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(tie + ((NC * TC + 1) & ~1));   // [NC] truth sums | [NC] token sums

    const uint32_t rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32*)(lds + (int64_t)wid * a.wave_lds_words));
    const uint32_t img_bytes = 64u * PS * 16u;
    const uint32_t tru_off = img_bytes;                               // (with tokens the ONE image holds the votes, then the tokens of a step)
    uint32_t off[QMAX];
    {
        uint32_t c = (uint32_t)lane / PS, k = (uint32_t)lane - c * PS;
        const uint32_t dc = 64u / PS, dk = 64u - dc * PS;
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            off[q] = c * rowbytes + (k < RS ? k : RS - 1u) * 16u;
            c += dc; k += dk;
            if (k >= PS) { k -= PS; c += 1; }
        }
    }
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    // (wave-major over the grid: the steps of the last, partial round go to one wave of every CU before any CU gets a second one)
    const int64_t wave = sv_uniform64((int64_t)wid * gridDim.x + blockIdx.x);
    const int64_t nsteps = (a.P + 63) / 64;
    const int64_t total_bytes = a.P * (int64_t)rowbytes;
    // the copy of step st's 64 rows -- its votes, or (tok) its tokens -- into this wave's image
    auto issue = [&](int64_t st, bool tok = false) {
        const int64_t byte0 = st * 64 * (int64_t)rowbytes;
        const int64_t rem = total_bytes - byte0 - 16;
        const uint32_t lim = rem > 0x7fffffffll ? 0x7fffffffu : (uint32_t)rem;
        const char* g = reinterpret_cast<const char*>(tok ? a.tokens : a.answers) + byte0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            if ((uint32_t)q < PS) {
                const uint32_t o = off[q] < lim ? off[q] : lim;
                sv_dma16(g, o, rbase + (uint32_t)q * 1024u);
            }
        }
    };
    auto issue_truth = [&](int64_t st) {
        const int64_t left = a.P - st * 64;
        const uint32_t live_rows = left > 64 ? 64u : (uint32_t)left;
        const int32_t* tp = reinterpret_cast<const int32_t*>(sv_uniform64((int64_t)(uintptr_t)(a.truth + st * 64)));
        sv_dma4(tp, ((uint32_t)lane < live_rows ? (uint32_t)lane : 0u) * 4u, rbase + tru_off);
    };

    int64_t st = wave;
    int32_t cb[NC + 1];
    SP_WALL_START();
    // (tie and acc are adjacent: [NC][TC] words rounded to an even count, then 2 NC 64-bit sums)
    if (!sort_prefix_setup_classes<NV, NC>(a, cbeg, ordl, tie, ((NC * TC + 1) & ~1) + 4 * NC, cb)) return;
    SP_WALL(2);
    // (the load of n_valid has returned -- the verdict needed it --: nothing the compiler would wait for with vmcnt(0) follows the copy.  Starting the copy
    // BEFORE the classes for a list the caller has promised was measured in round 6: the wall-clock marks of one launch move by 1.3 us, its hipEvent time does
    // not, and launches of >= 2 rounds lose 1-2 us -- every wave's n_valid then queues behind 35 MB of images: profiles/r06_prefix_fixed_cost_ab.log)
    if (st < nsteps) { issue(st); issue_truth(st); }
    // votes the longest budget sees (the domain check looks no further)
    int32_t nmax = 0;
#pragma unroll
    for (int c = 1; c < CF; ++c) if (cb[c + 1] > cb[c]) nmax = 1 << (c - 1);
    if (cb[CF + 1] > cb[CF]) nmax = N;

    // per class, per lane: truth votes (and tokens) of its problems; uniform: hits with one mode
    uint32_t tcs[NC];
    uint32_t h1[NC];
    long long toks[TOK ? NC : 1];
#pragma unroll
    for (int c = 0; c < NC; ++c) { tcs[c] = 0; h1[c] = 0; if (TOK) toks[c] = 0; }
    uint32_t bad = 0;
    // the previous step's records, packed: max_count | truth_count << 7 | n_modes << 14 | min_mode << 21 | hit << 31
    uint32_t D[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) D[c] = 0;
    int64_t Dp0 = 0;
    uint32_t Dlive = 0;
    // The [64][B] records of the previous step leave through this wave's image, between the step's rows having been read and the next copy
    // being started: every lane writes its B records where they lie in memory (cells[p][b], 16 B apart), then the table goes out in pieces of
    // 1 KiB of consecutive addresses.  (Written straight from the lanes -- 16 B every 16 B bytes apart -- a step spent 12-20 000 cycles
    // ISSUING its B stores: profiles/r05_sort_prefix_timeline.log.)  Budget lists longer than the image (PS KiB) are written directly.
    auto flush_records = [&]() {
        if (!a.cells || Dlive == 0) return;
        const bool staged = (uint32_t)B <= PS;
        uint4* const rowp = reinterpret_cast<uint4*>(a.cells) + (Dp0 + lane) * (int64_t)B;
        const uint32_t lrow = rbase + (uint32_t)lane * (uint32_t)B * 16u;
        const bool wr = (uint32_t)lane < Dlive;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (cb[c + 1] > cb[c]) {
                const uint32_t d = D[c];
                const scv_v4u rec = c == 0 ? scv_v4u{0u, 0u, 0xffff0000u, 0u}
                                           : scv_v4u{d & 0x7fu, (d >> 7) & 0x7fu, ((d >> 14) & 0x7fu) | (((d >> 21) & 0x3ffu) << 16), d >> 31};
                for (int32_t j = cb[c]; j < cb[c + 1]; ++j) {
                    const int32_t b = __builtin_amdgcn_readfirstlane(ordl[j]);
                    if (staged) *reinterpret_cast<lds_v4u*>((uintptr_t)(lrow + (uint32_t)b * 16u)) = rec;
                    else if (wr) __builtin_nontemporal_store(rec, reinterpret_cast<scv_v4u*>(rowp) + b);
                }
            }
        }
        if (staged) {
            scv_v4u* const out = reinterpret_cast<scv_v4u*>(a.cells) + Dp0 * (int64_t)B;
            const uint32_t nrec = Dlive * (uint32_t)B;
            for (int32_t i = 0; i < B; ++i) {
                const uint32_t k = (uint32_t)i * 64u + (uint32_t)lane;
                const scv_v4u rec = *reinterpret_cast<lds_v4u*>((uintptr_t)(rbase + k * 16u));
                if (k < nrec) __builtin_nontemporal_store(rec, out + k);
            }
        }
        Dlive = 0;
    };

#ifdef SCV_SP_TIMELINE
    unsigned long long tl[7] = {0, 0, 0, 0, 0, 0, 0}, tl_steps = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    asm volatile("; SCV_STEP_LOOP" ::: "memory");                    // (tests/test_abi_symbols.py counts the vmcnt waits behind this line)
    for (; st < nsteps; st += nwaves) {
        SP_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this step's images and truths have landed (and every older store)
        SP_STAMP(0);
        if (st == wave) SP_WALL(3);
        const int64_t left = a.P - st * 64;
        const uint32_t live_rows = left >= 64 ? 64u : (uint32_t)left;
        const bool live = (uint32_t)lane < live_rows;
        const int32_t trj = (int32_t)*reinterpret_cast<lds_u32*>((uintptr_t)(rbase + tru_off + (uint32_t)lane * 4u));
        const uint32_t tcmp = (trj >= 0 && trj < kBins) ? (uint32_t)trj : 0x7fffu;
        const uint32_t ra = rbase + (uint32_t)lane * (PS * 16u);
        uint32_t w[NV];
#pragma unroll
        for (int k = 0; k < RSM / 2; ++k) {
            const uint32_t k0 = (uint32_t)k < RS ? (uint32_t)k : RS - 1u, k1 = (uint32_t)(k + RSM / 2) < RS ? (uint32_t)(k + RSM / 2) : RS - 1u;
            const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k0));
            const scv_v4u h = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k1));
            w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
            w[NP + 4 * k] = h.x; w[NP + 4 * k + 1] = h.y; w[NP + 4 * k + 2] = h.z; w[NP + 4 * k + 3] = h.w;
        }
        uint32_t R[NP];
#pragma unroll
        for (int r = 0; r < NP; ++r) R[r] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(w[r], w[r + NP]));
        uint32_t orv = 0;
#pragma unroll
        for (int r = 0; r < NP; ++r) orv |= R[r];
        if (__any((orv & 0xfc00fc00u) != 0u)) {                      // (rare: o1.py:140 int(extracted_answer) is unbounded; the extractor maps it into the bins)
            int32_t nm = nmax;
            asm volatile("" : "+s"(nm));                             // (nmax laundered: hoisted out of this rare branch, its 64 uniform "i < nmax" masks are 128 SGPRs held -- spilled -- across the whole step loop)
            const int32_t seen = live ? nm : 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                bad |= i < seen ? w[i] : 0u;
                w[i] = w[i] < 1023u ? w[i] : 1023u;
            }
#pragma unroll
            for (int r = 0; r < NP; ++r) R[r] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(w[r], w[r + NP]));
        }
        if (N != NV) {                                               // slots behind the row: distinct sentinels behind every vote
            const uint32_t n2 = (uint32_t)N | ((uint32_t)N << 16);
#pragma unroll
            for (int r = 0; r < NP; ++r)
                R[r] = sv_sentinel(R[r], n2, (uint32_t)(r + 1) | ((uint32_t)(r + NP + 1) << 16),
                                   (0x8000u | (uint32_t)r) | ((0x8000u | (uint32_t)(r + NP)) << 16));
        }
        // truth votes among the first 2^j votes (index order, before the sort), j = 0 .. LG: one byte each
        uint32_t tcp[2] = {0u, 0u};
        {
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                run += w[i] == tcmp ? 1u : 0u;
                const int idx1 = i + 1;
                if ((idx1 & (idx1 - 1)) == 0) {
                    const int j = __builtin_ctz((unsigned)idx1);
                    tcp[j / 4] |= run << (8 * (j % 4));
                    tcs[1 + j] += live ? run : 0u;                   // (a class without budgets: never read)
                }
            }
        }
        SP_STAMP(1);
        if (st == wave) SP_WALL(4);
        // the previous step's records leave now: a whole sort lies between these stores and the next wait
        flush_records();
        SP_STAMP(2);
        const bool have_next = st + nwaves < nsteps;
        constexpr bool CAN_SPREAD = !TOK;
        const bool spread = CAN_SPREAD && have_next;
        const int64_t nbyte0 = (st + nwaves) * 64 * (int64_t)rowbytes;
        const int64_t nrem = total_bytes - nbyte0 - 16;
        const uint32_t nlim = nrem > 0x7fffffffll ? 0x7fffffffu : (nrem < 0 ? 0u : (uint32_t)nrem);
        const char* const ng = reinterpret_cast<const char*>(a.answers) + nbyte0;
        if constexpr (TOK) issue(st, true);                          // this step's TOKENS follow its votes through the image while the votes are sorted
        else if (have_next) {
            if (spread) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue_truth(st + nwaves); }
            else { issue(st + nwaves); issue_truth(st + nwaves); }
        }
        constexpr int STEP = sv_sort_ticks<NP>() / QMAX > 0 ? sv_sort_ticks<NP>() / QMAX : 1;
        int ticks = 0;
        auto piece = [&](int q, uint32_t& dep) __attribute__((always_inline)) {
            if constexpr (CAN_SPREAD) {
                if ((uint32_t)q < PS) sv_dma16_pinned(ng, off[q] < nlim ? off[q] : nlim, (uint32_t)__builtin_amdgcn_readfirstlane((int)(rbase + (uint32_t)q * 1024u)), dep);
            }
        };
        auto tick = [&](uint32_t& reg) __attribute__((always_inline)) {
            if constexpr (CAN_SPREAD) {
                if (spread && ticks % STEP == 0 && ticks / STEP < QMAX) piece(ticks / STEP, reg);
            }
            ++ticks;
        };
        // record + counters of one class from its statistics (tc = truth votes of the class's prefix)
        auto close_class = [&](int c, uint32_t maxc, uint32_t n_modes, uint32_t mm, uint32_t tc) {
            const uint32_t hit = tc == maxc ? 1u : 0u;                // o1.py:206 (every class here has votes)
            D[c] = maxc | (tc << 7) | (n_modes << 14) | (mm << 21) | (hit << 31);
            h1[c] += (uint32_t)__builtin_popcountll(__ballot(live && hit && n_modes == 1u));
            if (live && hit && n_modes != 1u) atomicAdd(&tie[c * TC + (int32_t)n_modes], 1u);
        };
        if (cb[2] > cb[1]) close_class(1, 1u, 1u, R[0] & 0x3ffu, tcp[0] & 0xffu);      // the first vote alone
        auto hook = [&](auto m_tag) __attribute__((always_inline)) {
            constexpr int M = decltype(m_tag)::value;
            constexpr int c = 1 + sv_log2(M);
            SP_STAMP(3);
            if (cb[c + 1] > cb[c]) {
                const BlockStats s = sv_scan_block<M, NP>(R);
                close_class(c, s.max_run, s.at_max, s.min_at_max, (tcp[(c - 1) / 4] >> (8 * ((c - 1) % 4))) & 0xffu);
            }
            SP_STAMP(4);
        };
        sv_sort_phases<NP, 1>(R, tick, hook);
        sv_merge_halves<NP>(R, tick);
        if constexpr (CAN_SPREAD) {
            if (spread) {
#pragma unroll
                for (int q = (sv_sort_ticks<NP>() + STEP - 1) / STEP; q < QMAX; ++q) piece(q, R[0]);
            }
        }
        SP_STAMP(3);
        if (st == wave) SP_WALL(5);
        if constexpr (TOK) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tokens have landed
            int64_t* const ctok_row = a.cell_tokens ? a.cell_tokens + (st * 64 + lane) * (int64_t)B : nullptr;
            {
                // running token sum in index order: the sums of the first 1, 2, 4 ... NV / 2 tokens and of all N.  (The step's tokens were
                // copied into the image while its votes were sorted.)  Every row is read before anything is staged: the [64][B] sums leave
                // like the records -- through the same image, in memory order.
                long long snap[LG + 2];
                long long run = 0;
#pragma unroll
                for (int k = 0; k < RSM; ++k) {
                    if ((uint32_t)k < RS) {
                        const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k));
                        const int32_t y[4] = {(int32_t)q.x, (int32_t)q.y, (int32_t)q.z, (int32_t)q.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            run += (long long)y[e];
                            const int idx1 = 4 * k + e + 1;               // tokens summed so far (a constant after unrolling)
                            if ((idx1 & (idx1 - 1)) == 0 && idx1 <= NP) snap[__builtin_ctz((unsigned)idx1)] = run;
                        }
                    }
                }
                snap[LG + 1] = run;
                const bool tstaged = (uint32_t)B <= PS;
                const uint32_t ltok = rbase + (uint32_t)lane * (uint32_t)B * 8u;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (cb[c + 1] > cb[c]) {
                        const long long v = c == 0 ? 0ll : snap[c - 1];
                        toks[c] += live ? v : 0ll;
                        if (a.cell_tokens) {
                            for (int32_t j = cb[c]; j < cb[c + 1]; ++j) {
                                const int32_t b = __builtin_amdgcn_readfirstlane(ordl[j]);
                                if (tstaged) *reinterpret_cast<lds_v2u*>((uintptr_t)(ltok + (uint32_t)b * 8u)) = scv_v2u{(uint32_t)(unsigned long long)v, (uint32_t)((unsigned long long)v >> 32)};
                                else if (live) ctok_row[b] = v;
                            }
                        }
                    }
                }
                if (a.cell_tokens && tstaged) {
                    char* const out = reinterpret_cast<char*>(a.cell_tokens + st * 64 * (int64_t)B);
                    const uint32_t ntok = live_rows * (uint32_t)B;
                    for (int32_t i = 0; 2 * 64 * i < 64 * B; ++i) {       // 16 bytes = two sums per lane and piece
                        const uint32_t k = (uint32_t)i * 64u + (uint32_t)lane;
                        const scv_v4u two = *reinterpret_cast<lds_v4u*>((uintptr_t)(rbase + k * 16u));
                        if (2u * k + 1u < ntok) __builtin_nontemporal_store(two, reinterpret_cast<scv_v4u*>(out) + k);
                        else if (2u * k < ntok) *reinterpret_cast<scv_v2u*>(out + (size_t)k * 16u) = scv_v2u{two.x, two.y};
                    }
                }
            }
            if (have_next) { issue(st + nwaves); issue_truth(st + nwaves); }
        }
        if (cb[CF + 1] > cb[CF]) {
            const SortedStats s = sv_scan<NP>(R, tcmp | (tcmp << 16));
            const uint32_t n_modes = s.at_max - ((NV == 64 && s.max_run == 1u) ? (uint32_t)(NV - N) : 0u);
            tcs[CF] += live ? s.truth_votes : 0u;
            close_class(CF, s.max_run, n_modes, s.min_at_max, s.truth_votes);
        }
        Dp0 = st * 64;
        Dlive = live_rows;
        SP_STAMP(5);
        if (st == wave) SP_WALL(6);
#ifdef SCV_SP_TIMELINE
        ++tl_steps;
#endif
    }
    SP_WALL(7);
#ifdef SCV_SP_TIMELINE
    if (lane == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&scv_sp_timeline[i], tl[i]);
        atomicAdd(&scv_sp_timeline[7], tl_steps);
    }
#endif
    if (!SCV_RECORDS_LAST) flush_records();
#ifdef SCV_SP_WALL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SP_WALL(8);
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    const bool counters = a.tie_hits || a.truth_sum || (TOK && a.token_sum);
    if (counters) {
#pragma unroll
        for (int c = 1; c < NC; ++c) {
            if (cb[c + 1] > cb[c]) {
                const long long ts = wave_sum_i64((long long)tcs[c]);
                long long tk = 0;
                if (TOK) tk = wave_sum_i64(toks[c]);
                if (lane == 0) {
                    if (h1[c]) atomicAdd(&tie[c * TC + 1], h1[c]);
                    if (ts) atomicAdd(&acc[c], (unsigned long long)ts);
                    if (TOK && tk) atomicAdd(&acc[NC + c], (unsigned long long)tk);
                }
            }
        }
        __syncthreads();
        sort_prefix_hand_out_counters<NC, TC, TOK>(a, cbeg, ordl, tie, acc);
    }
    if (SCV_RECORDS_LAST) flush_records();
#ifdef SCV_SP_WALL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SP_WALL(9);
}


// ---- pools of 68 .. 128 votes: the reference's largest (o1.py:266-276: T = 2^18 -> N = 128) ------------------------------------------------
//
// scv_sort_prefix2: one lane per problem, the row in TWO halves through the same 17 KiB image (a whole 128-vote row per lane would be a 33 KiB
// image: four waves per CU).  Half A = votes 0 .. 63: the 64-vote machinery above, unchanged -- its merge phases give maj@2 .. 32, its finished
// sort maj@64 (packed scan); half B = votes 64 .. N - 1 is copied while A is sorted, sorted by the same network, and the two sorted halves are
// joined by a bitonic merge ACROSS the two register files (flip stage: lo(A[r]) against hi(B[31 - r]), hi(A[r]) against lo(B[31 - r]); then
// each file is a bitonic sequence of 64: the stage between the halves of a register, five lockstep stages).  maj@N is a run scan of the 128
// sorted values in plain 32-bit arithmetic (run lengths up to 128 do not fit the packed scan's 6-bit field): running (length, best, count, value),
// 10 VALU per vote, no keys kept.  Per step: phase A = wait, rows A -> registers, the previous step's records leave through the image, copy B,
// sort A with the block scans, scan 64; phase B = wait, rows B -> registers, copy the next A, sort B, merge, scan 128.  Classes: 0 | 1 + j for 2^j votes, j = 0 .. 6 | 8 = all N votes.
// A launch of ONE step per wave takes ~27 us (two copies, two sorts, a merge and a 128-vote dependent scan in sequence; scv_prefix_pool: 13 us):
// the host takes this kernel from 57 344 pools (6.6e4: 30 against 35 us, 2e5: 48 .. 63 against 82, 8e5: 168 against 297).  With tokens (TOK): the
// token rows are summed in token steps of their own behind the sort steps -- the waves the sort's partial last round leaves idle take them first (below); a
// second image for the tokens (four waves per CU) had measured 114 us at 2e5 pools in round 5, a token kernel of its own behind this one 91, the token steps 69.
template <int NP>
__device__ __forceinline__ void sv_flip_files(uint32_t (&A)[NP], uint32_t (&Bv)[NP]) {
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        const uint32_t t = __builtin_amdgcn_alignbit(Bv[NP - 1 - r], Bv[NP - 1 - r], 16);
        const uint32_t mn = pk_min_c(A[r], t), mx = pk_max_c(A[r], t);
        A[r] = mn;
        Bv[NP - 1 - r] = __builtin_amdgcn_alignbit(mx, mx, 16);
    }
}
// a register file whose 2 NP elements (element i = half i / NP of X[i % NP]) form a bitonic sequence -> ascending
template <int NP>
__device__ __forceinline__ void sv_merge_bitonic_file(uint32_t (&X)[NP]) {
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        const uint32_t t = __builtin_amdgcn_alignbit(X[r], X[r], 16);
        const uint32_t mn = pk_min_c(X[r], t), mx = pk_max_c(X[r], t);
        X[r] = (mn & 0xffffu) | (mx & 0xffff0000u);
    }
#pragma unroll
    for (int j = NP >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int l = r ^ j;
            if (l > r) sv_ce(X[r], X[l]);
        }
    }
}
// statistics.multimode of the 4 NP sorted values lo(A[0..]), hi(A[0..]), lo(B[0..]), hi(B[0..]) in one pass: a run's length grows until the value
// changes; the first run to reach a new maximum is the smallest such value (ascending order), later runs that reach it are counted
template <int NP>
__device__ __forceinline__ BlockStats sv_scan_files(const uint32_t (&A)[NP], const uint32_t (&Bv)[NP]) {
    // (length, value) of the best run as ONE key, length << 16 | ~value: its maximum is the longest run with the smallest value (the values ascend).  A
    // select `minv = len > best ? x : minv` per element is off the dependent chain: the compiler sank all 128 of them behind the loop and kept their
    // masks -- 256 SGPRs, spilled to VGPR lanes and read back: 168 v_writelane + as many v_readlane and their hazard s_nops per step.
    uint32_t prev = 0xffffffffu, len = 0, best = 0, cnt = 0, bkey = 0;
    auto feed = [&](uint32_t x) {
        len = x == prev ? len + 1u : 1u;
        prev = x;
        const uint32_t cnt_eq = cnt + (len == best ? 1u : 0u);
        cnt = len > best ? 1u : cnt_eq;
        best = best > len ? best : len;
        const uint32_t key = (len << 16) | (x ^ 0xffffu);
        bkey = bkey > key ? bkey : key;
        __builtin_amdgcn_sched_barrier(0);                           // (element by element: nothing of the later elements is worked out ahead into registers the kernel does not have)
    };
#pragma unroll
    for (int r = 0; r < NP; ++r) feed(A[r] & 0xffffu);
#pragma unroll
    for (int r = 0; r < NP; ++r) feed(A[r] >> 16);
#pragma unroll
    for (int r = 0; r < NP; ++r) feed(Bv[r] & 0xffffu);
#pragma unroll
    for (int r = 0; r < NP; ++r) feed(Bv[r] >> 16);
    return BlockStats{best, cnt, (bkey & 0xffffu) ^ 0xffffu};
}

// ... of the M sorted values in the LOW halves of R[0 .. M - 1], the same way (no key per element held: the 128-vote kernel has no registers for them)
template <int M, int NP>
__device__ __forceinline__ BlockStats sv_scan_block_running(const uint32_t (&R)[NP]) {
    uint32_t prev = 0xffffffffu, len = 0, best = 0, cnt = 0, bkey = 0;   // (the best run as one key: see sv_scan_files)
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const uint32_t x = R[i] & 0xffffu;
        len = x == prev ? len + 1u : 1u;
        prev = x;
        const uint32_t cnt_eq = cnt + (len == best ? 1u : 0u);
        cnt = len > best ? 1u : cnt_eq;
        best = best > len ? best : len;
        const uint32_t key = (len << 16) | (x ^ 0xffffu);
        bkey = bkey > key ? bkey : key;
        __builtin_amdgcn_sched_barrier(0);
    }
    return BlockStats{best, cnt, (bkey & 0xffffu) ^ 0xffffu};
}

constexpr int sort_prefix2_threads() { return 512; }

// Host contract: 64 < N <= 128, N % 4 == 0, 16-byte aligned bases, B <= kMaxSortedB, every budget 0, a power of two <= 64 or >= N (checked
// here: a list that is not leaves the launch to the kernel queued behind it), a.tokens / a.cell_tokens / a.token_sum only for TOK; a.wave_lds_words = 64 * 17 * 4 + 64.
template <bool TOK>
__global__ __launch_bounds__(sort_prefix2_threads()) void scv_sort_prefix2(const AggArgs a) {
    constexpr int NV = 128, NH = 64, NP = 32, RSH = 16;             // votes per lane; per half; packed registers per half; 16-byte slots per half row
    constexpr uint32_t PS = 17u;                                     // slots of a padded half row in the image
    constexpr int TC = NV + 1;
    constexpr int NC = 9, CF = 8;                                    // classes 0 | 1 .. 7 (1 .. 64 votes) | 8 (all N)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, T = (int)blockDim.x, NW = T >> 6;
    const int32_t N = (int32_t)a.N, B = a.B;
    const uint32_t nB = (uint32_t)N - (uint32_t)NH;                  // votes of half B (4 .. 64)
    const uint32_t RSB = nB >> 2;                                    // its slots
    const uint32_t rowbytes = (uint32_t)N * 4u;
    int32_t* cbeg = reinterpret_cast<int32_t*>(lds + (int64_t)NW * a.wave_lds_words);
    int32_t* ordl = cbeg + 16;
    uint32_t* tie = reinterpret_cast<uint32_t*>(ordl + ((B + 3) & ~3));                  // [NC][TC]
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(tie + ((NC * TC + 1) & ~1));   // [NC] truth sums | [NC] token sums

    const uint32_t rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32*)(lds + (int64_t)wid * a.wave_lds_words));
    constexpr uint32_t img_bytes = 64u * PS * 16u;
    const uint32_t tru_off = img_bytes;
    // slot s = 64 q + lane of the image is chunk k = s % 17 of row c = s / 17; s += 64 is (c, k) += (3, 13) with a carry: walked again by every
    // copy (17 registers of offsets held across the loop were the registers the kernel did not have)
    const uint32_t c0 = (uint32_t)lane / PS, k0 = (uint32_t)lane - c0 * PS;
    const int64_t nwaves = (int64_t)gridDim.x * NW;
    // (wave-major over the grid: the steps of the last, partial round go to one wave of every CU before any CU gets a second one)
    const int64_t wave = sv_uniform64((int64_t)wid * gridDim.x + blockIdx.x);
    const int64_t nsteps = (a.P + 63) / 64;
    const int64_t total_bytes = a.P * (int64_t)rowbytes;
    // the copy of half h (0: votes 0 .. 63, 1: votes 64 .. N - 1) of step st's 64 rows into this wave's image(s)
    auto issue_half = [&](int64_t st, int h, bool tok = false) {
        const int64_t byte0 = st * 64 * (int64_t)rowbytes;
        const int64_t rem = total_bytes - byte0 - 16;
        const uint32_t lim = rem > 0x7fffffffll ? 0x7fffffffu : (uint32_t)rem;
        const char* g = reinterpret_cast<const char*>(tok ? a.tokens : a.answers) + byte0;
        const uint32_t kmax = h ? RSB - 1u : (uint32_t)RSH - 1u, add = h ? 256u : 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the rows of the image's previous content are in registers, staged tables have left)
        uint32_t cr = c0 * rowbytes + add, kq = k0;
        asm volatile("" : "+v"(cr), "+v"(kq));                       // (walked again by EVERY copy: hoisted out of the step loop, the 2 x 17 offsets were spilled to scratch)
#pragma unroll
        for (int q = 0; q < (int)PS; ++q) {
            uint32_t o = cr + (kq < kmax ? kq : kmax) * 16u;
            o = o < lim ? o : lim;
            sv_dma16(g, o, rbase + (uint32_t)q * 1024u);
            kq += 64u - 3u * PS; cr += 3u * rowbytes;
            if (kq >= PS) { kq -= PS; cr += rowbytes; }
        }
    };
    auto issue_truth = [&](int64_t st) {
        const int64_t left = a.P - st * 64;
        const uint32_t live_rows = left > 64 ? 64u : (uint32_t)left;
        const int32_t* tp = reinterpret_cast<const int32_t*>(sv_uniform64((int64_t)(uintptr_t)(a.truth + st * 64)));
        sv_dma4(tp, ((uint32_t)lane < live_rows ? (uint32_t)lane : 0u) * 4u, rbase + tru_off);
    };

    int64_t st = wave;
    int32_t cb[NC + 1];
    if (!sort_prefix_setup_classes<NV, NC>(a, cbeg, ordl, tie, ((NC * TC + 1) & ~1) + 4 * NC, cb)) return;
    if (st < nsteps) { issue_half(st, 0); issue_truth(st); }
    int32_t nmax = 0;                                                // votes the longest budget sees (the domain check looks no further)
#pragma unroll
    for (int c = 1; c < CF; ++c) if (cb[c + 1] > cb[c]) nmax = 1 << (c - 1);
    if (cb[CF + 1] > cb[CF]) nmax = N;
    const bool want_full = cb[CF + 1] > cb[CF];
    const bool need_b = want_full;                                   // (no budget beyond 64 votes: half B is never read)

    uint32_t tcs[NC];
    uint32_t h1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { tcs[c] = 0; h1[c] = 0; }
    uint32_t bad = 0;
    // the previous step's records, packed: D[c] = max_count | truth_count << 8 | n_modes << 16 | hit << 31; min_mode (10 bits) of class c
    // in Dm[c / 3] at bit 10 (c % 3)
    uint32_t D[NC], Dm[3] = {0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < NC; ++c) D[c] = 0;
    int64_t Dp0 = 0;
    uint32_t Dlive = 0;
    const bool staged = (uint32_t)B <= PS;
    auto flush_records = [&]() {                                     // through the votes image, in memory order (see scv_sort_prefix)
        if (!a.cells || Dlive == 0) return;
        uint4* const rowp = reinterpret_cast<uint4*>(a.cells) + (Dp0 + lane) * (int64_t)B;
        const uint32_t lrow = rbase + (uint32_t)lane * (uint32_t)B * 16u;
        const bool wr = (uint32_t)lane < Dlive;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (cb[c + 1] > cb[c]) {
                const uint32_t d = D[c], mm = (Dm[c / 3] >> (10 * (c % 3))) & 0x3ffu;
                const scv_v4u rec = c == 0 ? scv_v4u{0u, 0u, 0xffff0000u, 0u}
                                           : scv_v4u{d & 0xffu, (d >> 8) & 0xffu, ((d >> 16) & 0xffu) | (mm << 16), d >> 31};
                for (int32_t j = cb[c]; j < cb[c + 1]; ++j) {
                    const int32_t b = __builtin_amdgcn_readfirstlane(ordl[j]);
                    if (staged) *reinterpret_cast<lds_v4u*>((uintptr_t)(lrow + (uint32_t)b * 16u)) = rec;
                    else if (wr) __builtin_nontemporal_store(rec, reinterpret_cast<scv_v4u*>(rowp) + b);
                }
            }
        }
        if (staged) {
            scv_v4u* const out = reinterpret_cast<scv_v4u*>(a.cells) + Dp0 * (int64_t)B;
            const uint32_t nrec = Dlive * (uint32_t)B;
            for (int32_t i = 0; i < B; ++i) {
                const uint32_t k = (uint32_t)i * 64u + (uint32_t)lane;
                const scv_v4u rec = *reinterpret_cast<lds_v4u*>((uintptr_t)(rbase + k * 16u));
                if (k < nrec) __builtin_nontemporal_store(rec, out + k);
            }
        }
        Dlive = 0;
    };

    asm volatile("; SCV_STEP_LOOP" ::: "memory");
    for (; st < nsteps; st += nwaves) {
        // (values every step works out again: hoisted out of the loop, what the compiler derives from them -- 16 slot conditions, 16 clamped offsets, the sentinel
        // constants -- are ~50 SGPRs the kernel does not have: spilled to VGPR lanes and read back inside the step)
        uint32_t RSB_l = RSB, nB_l = nB;
        asm volatile("" : "+s"(RSB_l), "+s"(nB_l));
        // ================================ phase A: votes 0 .. 63 ================================
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // half A and the truths have landed (and every older store)
        const int64_t left = a.P - st * 64;
        const uint32_t live_rows = left >= 64 ? 64u : (uint32_t)left;
        const bool live = (uint32_t)lane < live_rows;
        const int32_t trj = (int32_t)*reinterpret_cast<lds_u32*>((uintptr_t)(rbase + tru_off + (uint32_t)lane * 4u));
        const uint32_t tcmp = (trj >= 0 && trj < kBins) ? (uint32_t)trj : 0x7fffu;
        const uint32_t ra = rbase + (uint32_t)lane * (PS * 16u);
        uint32_t w[NH];
#pragma unroll
        for (int k = 0; k < RSH; ++k) {
            const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k));
            w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
        }
        flush_records();                                             // the previous step's records leave through the image: its rows are in registers
        if (need_b) issue_half(st, 1);                               // half B flies while A is sorted
        else if (st + nwaves < nsteps) { issue_half(st + nwaves, 0); issue_truth(st + nwaves); }
        uint32_t RA[NP];
#pragma unroll
        for (int r = 0; r < NP; ++r) RA[r] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(w[r], w[r + NP]));
        {
            uint32_t orv = 0;
#pragma unroll
            for (int r = 0; r < NP; ++r) orv |= RA[r];
            if (__any((orv & 0xfc00fc00u) != 0u)) {
                int32_t nm = nmax;
                asm volatile("" : "+s"(nm));
                const int32_t seen = live ? (nm < NH ? nm : NH) : 0;
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    bad |= i < seen ? w[i] : 0u;
                    w[i] = w[i] < 1023u ? w[i] : 1023u;
                }
#pragma unroll
                for (int r = 0; r < NP; ++r) RA[r] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(w[r], w[r + NP]));
            }
        }
        // truth votes among the first 2^j votes (index order), j = 0 .. 6: one byte each; tcA = among all 64
        uint32_t tcp[2] = {0u, 0u};
        uint32_t tcA = 0;
        {
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                run += w[i] == tcmp ? 1u : 0u;
                const int idx1 = i + 1;
                if ((idx1 & (idx1 - 1)) == 0) {
                    const int j = __builtin_ctz((unsigned)idx1);
                    tcp[j / 4] |= run << (8 * (j % 4));
                    tcs[1 + j] += live ? run : 0u;
                }
            }
            tcA = run;
        }
        Dm[0] = Dm[1] = Dm[2] = 0u;
        auto close_class = [&](int c, uint32_t maxc, uint32_t n_modes, uint32_t mm, uint32_t tc) {
            const uint32_t hit = tc == maxc ? 1u : 0u;                // o1.py:206 (every class here has votes)
            D[c] = maxc | (tc << 8) | (n_modes << 16) | (hit << 31);
            Dm[c / 3] |= mm << (10 * (c % 3));
            h1[c] += (uint32_t)__builtin_popcountll(__ballot(live && hit && n_modes == 1u));
            if (live && hit && n_modes != 1u) atomicAdd(&tie[c * TC + (int32_t)n_modes], 1u);
        };
        auto tc_of = [&](int c) -> uint32_t { return (tcp[(c - 1) / 4] >> (8 * ((c - 1) % 4))) & 0xffu; };
        if (cb[2] > cb[1]) close_class(1, 1u, 1u, RA[0] & 0x3ffu, tc_of(1));
        SvNoTick none;
        auto hook = [&](auto m_tag) __attribute__((always_inline)) {
            constexpr int M = decltype(m_tag)::value;
            constexpr int c = 1 + sv_log2(M);
            if (cb[c + 1] > cb[c]) {
                const BlockStats s = sv_scan_block_running<M, NP>(RA);
                close_class(c, s.max_run, s.at_max, s.min_at_max, tc_of(c));
            }
        };
        sv_sort_phases<NP, 1>(RA, none, hook);
        sv_merge_halves<NP>(RA, none);
        if (cb[8] > cb[7]) {                                         // the first 64 votes: the packed scan of the sorted half
            const SortedStats s = sv_scan<NP>(RA, 0x7fff7fffu);
            close_class(7, s.max_run, s.at_max, s.min_at_max, tc_of(7));
        }
        // ================================ phase B: votes 64 .. N - 1 ================================
        if (!need_b) { Dp0 = st * 64; Dlive = live_rows; continue; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // half B has landed
        uint32_t RB[NP];
        uint32_t tcB = 0;
        {
            uint32_t v[NH];
#pragma unroll
            for (int k = 0; k < RSH; ++k) {
                const uint32_t kk = (uint32_t)k < RSB_l ? (uint32_t)k : RSB_l - 1u;
                const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * kk));
                v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
            }
#pragma unroll
            for (int r = 0; r < NP; ++r) RB[r] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(v[r], v[r + NP]));
            uint32_t orv = 0;
#pragma unroll
            for (int r = 0; r < NP; ++r) orv |= RB[r];
            if (__any((orv & 0xfc00fc00u) != 0u)) {
                int32_t nm = nmax;
                asm volatile("" : "+s"(nm));
                const int32_t seen = live ? (nm - NH > 0 ? nm - NH : 0) : 0;
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    bad |= i < seen ? v[i] : 0u;
                    v[i] = v[i] < 1023u ? v[i] : 1023u;
                }
#pragma unroll
                for (int r = 0; r < NP; ++r) RB[r] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(v[r], v[r + NP]));
            }
            if (want_full) {                                         // (whole slots under a scalar branch: 64 uniform "i < nB_l" masks cost 128 SGPRs)
#pragma unroll
                for (int k = 0; k < RSH; ++k) {
                    if ((uint32_t)k < RSB_l) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) tcB += v[4 * k + e] == tcmp ? 1u : 0u;
                    }
                }
            }
        }
        // the image is free: the next step's half A is copied
        if (st + nwaves < nsteps) { issue_half(st + nwaves, 0); issue_truth(st + nwaves); }
        if (want_full) {
            if (nB_l != (uint32_t)NH) {                                // slots behind the row: distinct sentinels behind every vote
                const uint32_t n2 = nB_l | (nB_l << 16);
#pragma unroll
                for (int r = 0; r < NP; ++r)
                    RB[r] = sv_sentinel(RB[r], n2, (uint32_t)(r + 1) | ((uint32_t)(r + NP + 1) << 16),
                                        (0x8000u | (uint32_t)r) | ((0x8000u | (uint32_t)(r + NP)) << 16));
            }
            sv_sort<NP>(RB, none);
            sv_flip_files<NP>(RA, RB);
            sv_merge_bitonic_file<NP>(RA);
            sv_merge_bitonic_file<NP>(RB);
            const BlockStats s = sv_scan_files<NP>(RA, RB);
            // (sentinels are runs of length 1: they count as modes only when every vote is distinct, and come off again)
            const uint32_t n_modes = s.at_max - (s.max_run == 1u ? (uint32_t)NV - (uint32_t)N : 0u);
            const uint32_t tc = tcA + tcB;
            tcs[CF] += live ? tc : 0u;
            close_class(CF, s.max_run, n_modes, s.min_at_max, tc);
        }
        Dp0 = st * 64;
        Dlive = live_rows;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (TOK || !SCV_RECORDS_LAST) flush_records();                   // (the token steps need the image)
    if (bad > 1023u) atomicOr(a.err_flag, 1u);
    // ================================ the token sums (round 6) ================================
    // cell_tokens[p][b] = sum of tokens[p][0 .. min(n_valid[b], N) - 1] (o1.py:195 over the budget's prefix), token_sum[b] = its sum over the problems
    // (o1.py:240).  A token step = the 64 token rows of 64 problems, one lane per row, through the wave's image in the same two halves as the votes: a
    // running 64-bit sum in index order, snapshots behind 1, 2, 4 ... 64 tokens and behind all N.  The tokens of a sample never influence its vote, so
    // ANY wave can take ANY token step -- and the sort leaves waves idle: its last round is partial (2e5 pools: 3125 steps on 2048 waves, 971 waves
    // have nothing to sort while the others sort their second step).  Those waves take the token steps first, SCV_TOK_PASSES each; what is left is dealt
    // to all waves.  (Round 6 first read the token rows in a kernel of its own behind this one, scv_prefix_tokens: 62 + 27 us + a launch gap at 2e5 pools.)
    long long toks[TOK ? NC : 1];
    if constexpr (TOK) {
#pragma unroll
        for (int c = 0; c < NC; ++c) toks[c] = 0;
        if (a.cell_tokens || a.token_sum) {
            const int64_t Lw = nsteps % nwaves;                       // waves 0 .. Lw - 1 sort one step more than the others
            const int64_t Iw = Lw == 0 ? 0 : nwaves - Lw;             // waves without a step in the sort's last round
            const int64_t v = wave >= Lw ? wave - Lw : wave + Iw;     // those first
            const bool tstaged = (uint32_t)B <= PS;
            const uint32_t ra = rbase + (uint32_t)lane * (PS * 16u);
            const uint32_t ltok = rbase + (uint32_t)lane * (uint32_t)B * 8u;
            int pass = v < Iw ? 0 : SCV_TOK_PASSES;
            auto step_of = [&](int p) -> int64_t { return p < SCV_TOK_PASSES ? (int64_t)p * Iw + v : (int64_t)SCV_TOK_PASSES * Iw + (int64_t)(p - SCV_TOK_PASSES) * nwaves + v; };
            int64_t t = step_of(pass);
            if (t < nsteps) issue_half(t, 0, true);
            while (t < nsteps) {
                uint32_t RSB_t = RSB;
                asm volatile("" : "+s"(RSB_t));                      // (per step, as in the sort loop: not 16 hoisted slot conditions in SGPRs)
                const int64_t left = a.P - t * 64;
                const uint32_t live_rows = left >= 64 ? 64u : (uint32_t)left;
                const bool live = (uint32_t)lane < live_rows;
                long long snap[8];
                long long run = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tokens 0 .. 63 of the step's rows have landed
                {
                    int32_t y[NH];
#pragma unroll
                    for (int k = 0; k < RSH; ++k) {
                        const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k));
                        y[4 * k] = (int32_t)q.x; y[4 * k + 1] = (int32_t)q.y; y[4 * k + 2] = (int32_t)q.z; y[4 * k + 3] = (int32_t)q.w;
                    }
                    if (want_full) issue_half(t, 1, true);           // tokens 64 .. N - 1 fly while the first 64 are summed
#pragma unroll
                    for (int i = 0; i < NH; ++i) {
                        run += (long long)y[i];
                        const int idx1 = i + 1;
                        if ((idx1 & (idx1 - 1)) == 0) snap[__builtin_ctz((unsigned)idx1)] = run;
                    }
                }
                if (want_full) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int k = 0; k < RSH; ++k) {
                        if ((uint32_t)k < RSB_t) {                   // (whole slots under a scalar branch, as for the votes)
                            const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k));
                            run += (long long)(int32_t)q.x + (long long)(int32_t)q.y + (long long)(int32_t)q.z + (long long)(int32_t)q.w;
                        }
                    }
                }
                snap[7] = run;
                // the [64][B] sums leave like the records: through the image, in memory order (every row has been read)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                int64_t* const ctok_row = a.cell_tokens ? a.cell_tokens + (t * 64 + lane) * (int64_t)B : nullptr;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (cb[c + 1] > cb[c]) {
                        const long long sv = c == 0 ? 0ll : snap[c - 1];
                        toks[c] += live ? sv : 0ll;
                        if (a.cell_tokens) {
                            for (int32_t j = cb[c]; j < cb[c + 1]; ++j) {
                                const int32_t b = __builtin_amdgcn_readfirstlane(ordl[j]);
                                if (tstaged) *reinterpret_cast<lds_v2u*>((uintptr_t)(ltok + (uint32_t)b * 8u)) = scv_v2u{(uint32_t)(unsigned long long)sv, (uint32_t)((unsigned long long)sv >> 32)};
                                else if (live) ctok_row[b] = sv;
                            }
                        }
                    }
                }
                if (a.cell_tokens && tstaged) {
                    char* const out = reinterpret_cast<char*>(a.cell_tokens + t * 64 * (int64_t)B);
                    const uint32_t ntok = live_rows * (uint32_t)B;
                    for (int32_t i = 0; 2 * 64 * i < 64 * B; ++i) {   // 16 bytes = two sums per lane and piece
                        const uint32_t k = (uint32_t)i * 64u + (uint32_t)lane;
                        const scv_v4u two = *reinterpret_cast<lds_v4u*>((uintptr_t)(rbase + k * 16u));
                        if (2u * k + 1u < ntok) __builtin_nontemporal_store(two, reinterpret_cast<scv_v4u*>(out) + k);
                        else if (2u * k < ntok) *reinterpret_cast<scv_v2u*>(out + (size_t)k * 16u) = scv_v2u{two.x, two.y};
                    }
                }
                ++pass;
                t = step_of(pass);
                if (t < nsteps) issue_half(t, 0, true);
            }
        }
    }
    const bool counters = a.tie_hits || a.truth_sum || (TOK && a.token_sum);
    if (counters) {
#pragma unroll
        for (int c = 1; c < NC; ++c) {
            if (cb[c + 1] > cb[c]) {
                const long long ts = wave_sum_i64((long long)tcs[c]);
                long long tk = 0;
                if (TOK) tk = wave_sum_i64(toks[c]);
                if (lane == 0) {
                    if (h1[c]) atomicAdd(&tie[c * TC + 1], h1[c]);
                    if (ts) atomicAdd(&acc[c], (unsigned long long)ts);
                    if (TOK && tk) atomicAdd(&acc[NC + c], (unsigned long long)tk);
                }
            }
        }
        __syncthreads();
        sort_prefix_hand_out_counters<NC, TC, TOK>(a, cbeg, ordl, tie, acc);
    }
    if (!TOK && SCV_RECORDS_LAST) flush_records();
}

}  // namespace scv
