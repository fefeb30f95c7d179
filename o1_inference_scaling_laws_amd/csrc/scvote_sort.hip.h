// scvote_sort.hip.h -- gfx950 (CDNA4 / MI355X): short cells, ONE LANE PER CELL, rows staged through LDS by LDS-DMA and
// SORTED in registers (scv_sort_cells).  The reference's own range: N = 1 ... 128 samples per cell (o1.py:267,276).
//
// Why another kernel for 8 <= N <= 64 (PMC of its predecessors: profiles/r02_regimes_pmc_register_kernels.md, r03_*):
//  * scv_reg_cells<16, 1, 4> (33 <= N <= 64) spreads a cell over 16 lanes: 4 votes per lane pay for two group reductions, the
//    pivot merge, the record and ~30 instructions of control -- ~36 VALU per vote with the LDS pipe half busy: 2.7-3.0 TB/s.
//  * scv_lane_cells<32> (N <= 32) has the right shape (a lane owns a cell, no cross-lane traffic) but every lane reads its own
//    128-byte row with eight 16-byte loads: a wave instruction touches 64 different cache lines, eight times each: 1.7 TB/s.
// Here the wave's 64 rows (one contiguous block of 64 * N * 4 bytes) are copied HBM -> LDS by global_load_lds_dwordx4: every
// instruction moves 1 KiB of consecutive addresses, no VGPR is staged, and the copy of the NEXT block is in flight while the
// current cells are counted (a block's counting time is several memory latencies, so one LDS buffer per wave is enough: the
// copy is issued as soon as the rows have been read into registers).  The LDS image is the block with one 16-byte pad slot
// per row when the row has an even number of slots: a lane's ds_read_b128 of its own row is then conflict-free (row stride
// odd in 16-byte slots).  The pad is made by the SOURCE addresses (the LDS side of the DMA is lane-linear).
//
// Counting (o1.py:181-195 + statistics.multimode + o1.py:204-213) without LDS, without cross-lane traffic, without compares:
// votes are 10-bit values, so two fit a register and v_pk_min_u16 / v_pk_max_u16 run TWO compare-exchanges per instruction
// pair.  Element i of the cell sits in half i / NP of register i % NP (NP = NV / 2): an odd-even mergesort network on the NP
// registers sorts both halves in lockstep and one bitonic merge joins them; the halves meet in one stage only (5 instructions
// per register pair there).  NV = 64: 622 instructions for 64 votes.  A scan over the sorted registers then gives, again two
// elements per instruction, the 1-based index of the run start each element belongs to (running maximum of
// [x_i != x_i-1] * index), hence key_i = (NV - length of the run ending at i) << 10 | x_i: the SMALLEST key is the last element
// of the longest run with the smallest value -- max_count and min(multimode) from one packed minimum per register --,
// len(multimode) = #{ i : key_i <= smallest key | 0x3ff } (a maximal run reaches max_count exactly once, at its last element:
// no division), truth_count = #{ x_i == truth }: 13 instructions per register (round 3: 16, with a pass of its own for the
// candidates of min_mode).  Votes past the valid prefix (n_valid[b] < NV) become DISTINCT sentinels 0x8000 | i: they sort
// behind every vote, form runs of length 1, and their keys saturate (v_pk_mad_u16 clamp) above every vote's key -- except in
// the 64-vote shape when every vote is distinct (run length 1 is key field 63, and 63 << 10 | 1023 is the saturated key),
// where all of them are counted and taken off again.
//
// What bounds it (round 4: tools/valu_probe.hip, tools/sort_timeline.py, hbm_probe --dmawork; profiles/r04_valu_probe.log,
// r04_sort_timeline*.log, r04_hbm_probe_dmawork.log): packed 16-bit and three-operand VALU instructions issue every 4.4 cycles
// per SIMD (plain 32-bit add / xor: 2.2), so ~20 of them per vote cap the shapes near 7 TB/s; a wave spends 40-58 % of its cycles
// blocked ISSUING its LDS-DMA pieces (the CU's memory pipe serves every wave's pieces in turn) or waiting for them, which two to
// four waves per SIMD cover to ~70 % VALU occupancy.  One producer wave per workgroup that issues every piece (consumers wait
// on LDS flags, one or two buffers each) does not beat the waves copying for themselves on the same work (4.0-4.5 against 4.7
// TB/s at 16 KiB blocks, worse for smaller blocks): measured in the probe, not built here.
#pragma once

#include <utility>

#include "scvote_kernels.hip.h"
#include "scvote_sortnet.h"

namespace scv {

// Packed 16-bit arithmetic of the scan as inline asm: written with clang's vector types, instcombine recognises the 0 / 1
// arithmetic (min(x, 1) * c, 1 -sat x ...) as selects and the backend then emits 16-bit compares + v_cndmask (SDWA) + v_perm --
// three times the instructions and 60+ live SGPR-pair masks (spilled).  Each step of the scan is ONE asm statement: hipcc pads
// every asm result that the next instruction reads with an s_nop (it must assume a dst_sel forwarding hazard; full-dword
// v_pk_* results have none), so single-instruction statements cost a wait state per dependent pair.  Constants are SGPR
// operands (s_mov literals: scalar issue, not VALU).

// The sorting network itself is plain min / max: nothing for instcombine to "simplify", so it goes through the compiler
// (which schedules it and needs no hazard padding around it: every inline-asm statement costs an s_nop now and then).
typedef unsigned short sv_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_c(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(sv_u16x2, a), __builtin_bit_cast(sv_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_c(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(sv_u16x2, a), __builtin_bit_cast(sv_u16x2, b)));
}
// compare-exchange of both halves: the minima to `lo`, the maxima to `hi`
__device__ __forceinline__ void sv_ce(uint32_t& lo, uint32_t& hi) {
    const uint32_t mn = pk_min_c(lo, hi), mx = pk_max_c(lo, hi);
    lo = mn; hi = mx;
}
// the one stage where the halves meet: (lo(a), hi(b)) and (lo(b), hi(a)), minima to the low halves
__device__ __forceinline__ void sv_ce_cross(uint32_t& a, uint32_t& b) {
    const uint32_t t = __builtin_amdgcn_alignbit(b, b, 16);
    const uint32_t mn = pk_min_c(a, t), mx = pk_max_c(a, t);
    a = (mn & 0xffffu) | (mx & 0xffff0000u);
    b = __builtin_amdgcn_alignbit(mx, mn, 16);
}

// (the network generator is plain C++: csrc/scvote_sortnet.h, also compiled by the CPU test tests/test_sort_network.py)
template <int N>
struct SvNet { static constexpr SvNetwork<N> net = sv_make_network<N>(); };
struct SvNoTick { __device__ __forceinline__ void operator()(uint32_t&) const {} };
template <int NP, typename Tick, int... I>
__device__ __forceinline__ void sv_sort_halves(uint32_t (&R)[NP], Tick& tick, std::integer_sequence<int, I...>) {
    ((sv_ce(R[SvNet<NP>::net.a[I]], R[SvNet<NP>::net.b[I]]), tick(R[SvNet<NP>::net.a[I]])), ...);      // (every index is a constant expression: R stays in registers)
}
constexpr bool sv_pow2(int n) { return n >= 2 && (n & (n - 1)) == 0; }
template <int N>
struct SvValley { static constexpr SvNetwork<N> net = sv_make_valley_merge<N>(); };
// compare-exchanges (= calls of `tick`) of one sv_sort<NP>
template <int NP>
constexpr int sv_sort_ticks() {
    if (!sv_pow2(NP)) return SvNet<NP>::net.n + NP + SvValley<NP>::net.n;
    int stages = 0;
    for (int j = NP >> 1; j > 0; j >>= 1) ++stages;
    return SvNet<NP>::net.n + NP / 2 + stages * (NP / 2);
}
template <int NP, typename Tick, int... I>
__device__ __forceinline__ void sv_merge_valleys(uint32_t (&R)[NP], Tick& tick, std::integer_sequence<int, I...>) {
    ((sv_ce(R[SvValley<NP>::net.a[I]], R[SvValley<NP>::net.b[I]]), tick(R[SvValley<NP>::net.a[I]])), ...);
}
// lo = ~min(a, b), hi = max(a, b) of a register that holds lo = ~a, hi = b: swap the halves, complement, one packed maximum
__device__ __forceinline__ void sv_ce_halves(uint32_t& r) { r = pk_max_c(r, ~__builtin_amdgcn_alignbit(r, r, 16)); }
template <int NP, typename Tick, int... I>
__device__ __forceinline__ void sv_meet_halves(uint32_t (&R)[NP], Tick& tick, std::integer_sequence<int, I...>) {
    ((sv_ce_halves(R[I]), tick(R[I])), ...);
}

// Ascending sort of the 2 * NP 16-bit elements of R; element i = half i / NP of R[i % NP].  Both halves are sorted in lockstep by
// the odd-even mergesort network on the NP registers (any network whose exchanges all put the minimum on the lower wire runs on both
// halves at once), then merged by one bitonic merge: the flip stage is the only one where the halves meet.  `tick(reg)` is called after
// every compare-exchange with a register it has just written (the kernel spreads the next step's LDS-DMA pieces over the sort with it).
//
// NP not a power of two (round 4: the 48-vote shape, NP = 24 -- cells of 33 ... 48 votes used to sort 64 slots).  A bitonic merge on n
// wires can leave out the wires of a +inf padding only when the padded sequence is still bitonic, i.e. when the n wires hold a VALLEY
// (falling, then rising); the flip stage leaves a peak in the lower half.  So half 0 travels COMPLEMENTED (x ^ 0xffff: the lockstep
// minimum / maximum then sort it descending): after the lockstep network the 2 * NP elements are A falling, B rising -- one valley --, the
// first merge stage compares the two halves of each REGISTER (swap halves, complement, one packed maximum: lo = ~min(a, b),
// hi = max(a, b): three instructions per register instead of five per pair), and both halves are valleys in their stored form, so
// Lang's merge for arbitrary n (the power-of-two network minus every exchange that touches a pad wire: 52 instead of 80 exchanges at
// NP = 24) sorts them in lockstep.  Result after the final un-complement: half 0 FALLING in r (the NP smallest elements), half 1
// rising (the NP largest): in value order the elements run (0, NP-1) ... (0, 0), (1, 0) ... (1, NP-1) -- the halves meet at r = 0.
template <int NP, typename Tick>
__device__ __forceinline__ void sv_sort(uint32_t (&R)[NP], Tick& tick) {
    static_assert(NP >= 2, "packed registers");
    if constexpr (!sv_pow2(NP)) {
#pragma unroll
        for (int r = 0; r < NP; ++r) R[r] ^= 0xffffu;
        sv_sort_halves<NP>(R, tick, std::make_integer_sequence<int, SvNet<NP>::net.n>{});
        sv_meet_halves<NP>(R, tick, std::make_integer_sequence<int, NP>{});
        sv_merge_valleys<NP>(R, tick, std::make_integer_sequence<int, SvValley<NP>::net.n>{});
#pragma unroll
        for (int r = 0; r < NP; ++r) R[r] ^= 0xffffu;
    } else {
        sv_sort_halves<NP>(R, tick, std::make_integer_sequence<int, SvNet<NP>::net.n>{});
#pragma unroll
        for (int r = 0; r < NP / 2; ++r) { sv_ce_cross(R[r], R[NP - 1 - r]); tick(R[r]); }   // element (0, r) against (1, NP - 1 - r)
#pragma unroll
        for (int j = NP >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int l = r ^ j;
                if (l > r) { sv_ce(R[r], R[l]); tick(R[r]); }
            }
        }
    }
}

struct SortedStats { uint32_t max_run, at_max, min_at_max, truth_votes; };

// statistics.multimode on the sorted registers (see the header comment); tcmp2 = the truth in both halves (0x7fff7fff: none)
template <int NP>
__device__ __forceinline__ SortedStats sv_scan(const uint32_t (&R)[NP], uint32_t tcmp2) {
    static_assert(NP % 4 == 0, "the scan steps take four registers per asm statement");
    constexpr bool MEET = !sv_pow2(NP);
    uint32_t run[NP];
    uint32_t s = 0;
#define SV_IDX(r, d) ((uint32_t)((r) + (d)) | ((uint32_t)((r) + NP + (d)) << 16))   /* (index of element (0, r)) + d | the same for (1, r) */
#pragma unroll
    for (int r = 0; r < NP; r += 4) {
        // predecessor of element (h, r): (h, r - 1); of (0, 0): none (0xffff differs from every element); of (1, 0): (0, NP - 1) --
        // MEET (NP not a power of two: the halves meet at r = 0, see sv_sort): (0, 0), whose run -- the FIRST of half 0 -- it continues
        const uint32_t prev = r ? R[r - 1] : ((R[MEET ? 0 : NP - 1] << 16) | 0xffffu);
        // 1 where a run starts, times the 1-based index of the element; running maximum = index of the latest start
        uint32_t t0, t1, t2, t3, o0, o1, o2, o3;
        asm("v_xor_b32 %0, %8, %12\n\t"
            "v_xor_b32 %1, %9, %8\n\t"
            "v_xor_b32 %2, %10, %9\n\t"
            "v_xor_b32 %3, %11, %10\n\t"
            "v_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\t"
            "v_pk_min_u16 %1, %1, 1 op_sel_hi:[1,0]\n\t"
            "v_pk_min_u16 %2, %2, 1 op_sel_hi:[1,0]\n\t"
            "v_pk_min_u16 %3, %3, 1 op_sel_hi:[1,0]\n\t"
            "v_pk_mul_lo_u16 %0, %0, %14\n\t"
            "v_pk_mul_lo_u16 %1, %1, %15\n\t"
            "v_pk_mul_lo_u16 %2, %2, %16\n\t"
            "v_pk_mul_lo_u16 %3, %3, %17\n\t"
            "v_pk_max_u16 %4, %13, %0\n\t"
            "v_pk_max_u16 %5, %4, %1\n\t"
            "v_pk_max_u16 %6, %5, %2\n\t"
            "v_pk_max_u16 %7, %6, %3"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
            : "v"(R[r]), "v"(R[r + 1]), "v"(R[r + 2]), "v"(R[r + 3]), "v"(prev), "v"(s),
              "s"(SV_IDX(r, 1)), "s"(SV_IDX(r + 1, 1)), "s"(SV_IDX(r + 2, 1)), "s"(SV_IDX(r + 3, 1)));
        run[r] = o0; run[r + 1] = o1; run[r + 2] = o2; run[r + 3] = o3;
        s = o3;
    }
    uint32_t carry = s << 16;                           // a run that crosses from half 0 into half 1 started in half 0
    if constexpr (MEET) {
        // ... here it is half 0's FIRST run (f0 elements: those whose latest start is still index 1) that continues into half 1: an
        // element of half 1 that has seen no start of its own belongs to a run that started f0 elements before index NP + 1
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < NP; r += 4) {
            uint32_t t0, t1, t2, t3, ao;
            asm("v_pk_min_u16 %0, %5, 2 op_sel_hi:[1,0]\n\t"
                "v_pk_min_u16 %1, %6, 2 op_sel_hi:[1,0]\n\t"
                "v_pk_min_u16 %2, %7, 2 op_sel_hi:[1,0]\n\t"
                "v_pk_min_u16 %3, %8, 2 op_sel_hi:[1,0]\n\t"
                "v_pk_add_u16 %0, %0, %1\n\t"
                "v_pk_add_u16 %2, %2, %3\n\t"
                "v_pk_add_u16 %4, %9, %0\n\t"
                "v_pk_add_u16 %4, %4, %2"
                : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(ao)
                : "v"(run[r]), "v"(run[r + 1]), "v"(run[r + 2]), "v"(run[r + 3]), "v"(acc));
            acc = ao;
        }
        carry = ((acc & 0xffffu) + 1u - (uint32_t)NP) << 16;     // (acc.lo = 2 NP - f0) -> NP + 1 - f0
    }
    // key of an element = (NV - length of the run ending at it) << 10 | value, NV = 2 NP: the SMALLEST key belongs to the last element of
    // the longest run with the smallest value -- max_count and min(multimode) from one packed minimum per register, no second pass for
    // the candidates.  NV - length = start + (NV - 1 - index1) with the 1-based start of the run (its own half's, or the carry);
    // a sentinel (>= 0x8000) saturates its key (clamp).
    constexpr int NV = 2 * NP;
    uint32_t key[NP];
    uint32_t km = 0xffffffffu;
#define SV_KOFF(r) (((uint32_t)(NV - 2 - (r)) & 0xffffu) | (((uint32_t)(NV - 2 - (r) - NP) & 0xffffu) << 16))   /* NV - 1 - index1 of (0, r) | of (1, r) */
#pragma unroll
    for (int r = 0; r < NP; r += 4) {
        uint32_t k0, k1, k2, k3, mo;
        asm("v_pk_max_u16 %0, %5, %9\n\t"
            "v_pk_max_u16 %1, %6, %9\n\t"
            "v_pk_max_u16 %2, %7, %9\n\t"
            "v_pk_max_u16 %3, %8, %9\n\t"
            "v_pk_add_u16 %0, %0, %11\n\t"
            "v_pk_add_u16 %1, %1, %12\n\t"
            "v_pk_add_u16 %2, %2, %13\n\t"
            "v_pk_add_u16 %3, %3, %14\n\t"
            "v_pk_mad_u16 %0, %0, %15, %16 clamp\n\t"
            "v_pk_mad_u16 %1, %1, %15, %17 clamp\n\t"
            "v_pk_mad_u16 %2, %2, %15, %18 clamp\n\t"
            "v_pk_mad_u16 %3, %3, %15, %19 clamp\n\t"
            "v_pk_min_u16 %4, %10, %0\n\t"
            "v_pk_min_u16 %4, %4, %1\n\t"
            "v_pk_min_u16 %4, %4, %2\n\t"
            "v_pk_min_u16 %4, %4, %3"
            : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3), "=&v"(mo)
            : "v"(run[r]), "v"(run[r + 1]), "v"(run[r + 2]), "v"(run[r + 3]), "v"(carry), "v"(km),
              "s"(SV_KOFF(r)), "s"(SV_KOFF(r + 1)), "s"(SV_KOFF(r + 2)), "s"(SV_KOFF(r + 3)), "v"(0x04000400u),
              "v"(R[r]), "v"(R[r + 1]), "v"(R[r + 2]), "v"(R[r + 3]));
        key[r] = k0; key[r + 1] = k1; key[r + 2] = k2; key[r + 3] = k3;
        km = mo;
    }
#undef SV_KOFF
#undef SV_IDX
    SortedStats o;
    const uint32_t k1 = (km & 0xffffu) < (km >> 16) ? (km & 0xffffu) : (km >> 16);
    o.max_run = (uint32_t)NV - (k1 >> 10);
    o.min_at_max = k1 & 0x3ffu;
    const uint32_t thr = k1 | 0x3ffu, thr2 = thr | (thr << 16);
    uint32_t above = 0, tc = 0;
#pragma unroll
    for (int r = 0; r < NP; r += 2) {
        // 1 where the run ending here is shorter than the longest (key above every key of that length); 1 where the element equals
        // the truth; both summed as packed 0 / 1 by 32-bit three-operand adds (no carry between the halves: sums <= NP)
        uint32_t e0, e1, d0, d1, ao, to;
        asm("v_pk_sub_u16 %0, %6, %8 clamp\n\t"
            "v_pk_sub_u16 %1, %7, %8 clamp\n\t"
            "v_xor_b32 %2, %9, %11\n\t"
            "v_xor_b32 %3, %10, %11\n\t"
            "v_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\t"
            "v_pk_min_u16 %1, %1, 1 op_sel_hi:[1,0]\n\t"
            "v_pk_sub_u16 %2, 1, %2 op_sel_hi:[0,1] clamp\n\t"
            "v_pk_sub_u16 %3, 1, %3 op_sel_hi:[0,1] clamp\n\t"
            "v_add3_u32 %4, %12, %0, %1\n\t"
            "v_add3_u32 %5, %13, %2, %3"
            : "=&v"(e0), "=&v"(e1), "=&v"(d0), "=&v"(d1), "=&v"(ao), "=&v"(to)
            : "v"(key[r]), "v"(key[r + 1]), "v"(thr2), "v"(R[r]), "v"(R[r + 1]), "v"(tcmp2), "v"(above), "v"(tc));
        above = ao; tc = to;
    }
    o.at_max = 2u * NP - ((above & 0xffffu) + (above >> 16));
    o.truth_votes = (tc & 0xffffu) + (tc >> 16);
    return o;
}

// slots past the valid prefix become DISTINCT sentinels 0x8000 | i behind every vote, on the packed register, by arithmetic:
// 1 where the 1-based index exceeds n, times the sentinel, maximum with the slot
__device__ __forceinline__ uint32_t sv_sentinel(uint32_t x, uint32_t n2, uint32_t idx1, uint32_t sent) {
    uint32_t t, o;
    asm("v_pk_sub_u16 %0, %3, %4 clamp\n\t"
        "v_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_lo_u16 %0, %0, %5\n\t"
        "v_pk_max_u16 %1, %2, %0"
        : "=&v"(t), "=v"(o)
        : "v"(x), "s"(idx1), "v"(n2), "s"(sent));
    return o;
}

// (one LDS-DMA piece: lds_dma16 of scvote_kernels.hip.h)
__device__ __forceinline__ void sv_dma16(const void* gbase, uint32_t goff, uint32_t lds_dst) { lds_dma16(gbase, goff, lds_dst); }
// ... the same, pinned in the data flow of the sort: `dep` (a register the preceding compare-exchange wrote and a later one reads) is
// an in/out operand the statement does not touch, so the exchanges before it stay before and the ones after it stay after -- without it
// the compiler sinks the whole sort below the pieces (asm statements only keep their order among themselves)
__device__ __forceinline__ void sv_dma16_pinned(const void* gbase, uint32_t goff, uint32_t lds_dst, uint32_t& dep) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(dep) : "v"(goff), "s"(gbase), "s"(lds_dst) : "memory");
}
// ... and 64 lanes x 4 bytes (lane l's dword lands at lds_dst + 4 l): the per-lane gather of the cells' truth values
__device__ __forceinline__ void sv_dma4(const void* gbase, uint32_t goff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(goff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int64_t sv_uniform64(int64_t v) { return uniform64(v); }

constexpr int sort_cells_threads(int nv) { return nv >= 64 ? 512 : 1024; }

// NV: votes per lane (capacity of the shape: 8 / 16 / 24 / 32 / 40 / 48 / 56 / 64; NV / 2 a multiple of 4, NV <= 64: the run-length field of a key is 6 bits).  Host contract: N % 4 == 0, 4 <= N <= NV, 16-byte aligned
// bases, no pool rows; a.wave_lds_words = words of one wave's LDS region (64 * PS * 4, twice that with tokens, + 64 for the cells'
// truth values; PS = (N / 4) | 1 slots per padded row); the workgroup's LDS = regions | n_valid cache | tie classes | sums.
//
// LIN (rows that are not all 16-byte aligned: N % 4 != 0 or unaligned bases; the reference's N is arbitrary, o1.py:276): a block
// of 64 rows is still ONE contiguous run of bytes (64 * N * 4 is a multiple of 16, so every block has the base's misalignment):
// the image is the 16-byte aligned superset of the block, linear, and a lane reads its row with N ds_read_b32 at lane stride N
// words (conflict-free for odd N, 2-way for N = 2 mod 4).  Host contract: 1 <= N <= NV, a.wave_lds_words covers
// 64 * N * 4 + 16 bytes rounded up to whole KiB (twice with tokens: the token base has its own misalignment).
//
// Measured and not kept (profiles/r03_sort_cells_ab.log, r03_ab_n128.log; DESIGN_HISTORY.md 4): two blocks of 64 cells per step, two image
// buffers per wave with the copy two steps ahead, 32 resident waves per CU for the 8-vote shape, a 128-vote shape staged in two
// half-rows, the copy's pieces issued back to back instead of between the compare-exchanges.
#ifdef SCV_SORT_TIMELINE
// Measurement build only (tools/sort_timeline.py; never defined for the product library): shader cycles every wave spent in the phases of
// a step -- wait for the copy | rows LDS -> registers, packed | sort (+ the next copy's pieces) | scan | records and counters --, summed
// over all waves, plus the number of steps.  (s_memtime returns through lgkmcnt: each stamp also drains the wave's LDS queue.)
__device__ unsigned long long scv_sort_timeline[8];
#define SV_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_now = __builtin_readcyclecounter(); \
                         __builtin_amdgcn_sched_barrier(0); tl[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define SV_STAMP(i) do { } while (0)
#endif

template <int NV, bool TOK, bool LIN = false>
__global__ __launch_bounds__(sort_cells_threads(NV)) void scv_sort_cells(const AggArgs a) {
    constexpr int NP = NV / 2, RSM = NV / 4;
    constexpr int QMAX = RSM + 1;                                    // DMA pieces per step at the widest row (LIN: 64 * 4 N + 16 bytes)
    constexpr int TC = NV + 1;                                       // tie classes 0..NV
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, T = (int)blockDim.x, NW = T >> 6;
    const int32_t N = (int32_t)a.N, B = a.B;
    const uint32_t RS = (uint32_t)N >> 2, PS = RS | 1u;              // slots of a row, of a padded row
    const uint32_t rowbytes = (uint32_t)N * 4u;
    const uint32_t shv = LIN ? (uint32_t)(uintptr_t)a.answers & 15u : 0u;            // LIN: bytes between the aligned superset and the block
    const uint32_t sht = (LIN && TOK) ? (uint32_t)(uintptr_t)a.tokens & 15u : 0u;
    const uint32_t nq = LIN ? (64u * rowbytes + 16u + 1023u) >> 10 : PS;             // DMA pieces per step and stream
    uint32_t* nv_lds = lds + (int64_t)NW * a.wave_lds_words;
    const bool nv_cached = a.n_valid && B <= kMaxSortedB;
    uint32_t* tie = nv_lds + (nv_cached ? ((B + 3) & ~3) : 0);       // [B][TC]
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(tie + (((int64_t)B * TC + 1) & ~(int64_t)1));   // [B] truth sums | [B] token sums
    const bool counters = a.tie_hits || a.truth_sum || (TOK && a.token_sum);
    if (nv_cached)
        for (int i = tid; i < B; i += T) nv_lds[i] = (uint32_t)valid_len(a, i);
    if (counters) {
        for (int64_t i = tid; i < (int64_t)B * TC; i += T) tie[i] = 0;
        for (int i = tid; i < 2 * B; i += T) acc[i] = 0;
    }
    __syncthreads();

    // this wave's region: votes image [64 rows][PS slots], then (TOK) the tokens image, then the cells' truth values
    const uint32_t rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32*)(lds + (int64_t)wid * a.wave_lds_words));
    const uint32_t img_bytes = LIN ? nq * 1024u : 64u * PS * 16u;
    const uint32_t tru_off = img_bytes * (TOK ? 2u : 1u);
    // source offset of every slot this lane copies: slot s = 64 q + lane is chunk k = s % PS of row c = s / PS (the pad slot,
    // k == RS, repeats the row's last chunk)
    uint32_t off[QMAX];
    if (LIN) {
#pragma unroll
        for (int q = 0; q < QMAX; ++q) off[q] = (uint32_t)q * 1024u + (uint32_t)lane * 16u;
    } else {
        uint32_t c = (uint32_t)lane / PS, k = (uint32_t)lane - c * PS;   // one division; then s += 64 is (c, k) += (64 / PS, 64 % PS) with a carry
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
    constexpr int64_t SC = 64;                                       // cells per step
    const int64_t nsteps = (a.ncells + SC - 1) / SC;
    const int64_t total_bytes = a.ncells * (int64_t)rowbytes;
    auto issue = [&](int64_t st) {                                   // (wave-uniform) start the copy of step st into this wave's image
        const int64_t byte0 = st * SC * (int64_t)rowbytes;
        // last 16-byte chunk that holds bytes of the tensor, relative to the (aligned superset of the) block
        const int64_t rem = LIN ? ((total_bytes - byte0 + shv - 1) & ~(int64_t)15) : total_bytes - byte0 - 16;
        const int64_t remt = (LIN && TOK) ? ((total_bytes - byte0 + sht - 1) & ~(int64_t)15) : rem;
        const uint32_t lim = rem > 0x7fffffffll ? 0x7fffffffu : (uint32_t)rem;
        const uint32_t limt = remt > 0x7fffffffll ? 0x7fffffffu : (uint32_t)remt;
        const char* g = reinterpret_cast<const char*>(a.answers) + byte0 - shv;
        const char* gt = TOK ? reinterpret_cast<const char*>(a.tokens) + byte0 - sht : nullptr;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the rows of the previous step have left the region
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            if ((uint32_t)q < nq) {
                const uint32_t o = off[q] < lim ? off[q] : lim;      // (slots past the last cell re-read the tensor's last chunk)
                sv_dma16(g, o, rbase + (uint32_t)q * 1024u);
                if (TOK) {
                    const uint32_t ot = off[q] < limt ? off[q] : limt;
                    sv_dma16(gt, ot, rbase + img_bytes + (uint32_t)q * 1024u);
                }
            }
        }
    };

    // Walkers.  The step's first cell c0 (problem p0, budget b0) advances on the scalar unit; this lane's cell is c0 + lane =
    // c0 + lq * B + lr: its budget and its problem relative to p0 cost three 32-bit VALU operations per step.
    const int64_t stride = nwaves * SC;
    const int64_t dp = sv_uniform64(stride / B);                     // (64-bit divisions run on the VALU: back to SGPRs)
    const int32_t db = (int32_t)(stride - dp * B);
    int64_t c0 = wave * SC;
    int64_t p0 = sv_uniform64(c0 / B);
    int32_t b0 = (int32_t)(c0 - p0 * B);
    const uint32_t lq = (uint32_t)lane / (uint32_t)B, lr = (uint32_t)lane - lq * (uint32_t)B;
    auto slot_budget = [&](uint32_t& prel) -> int32_t {              // budget of this lane's cell in the current step; prel = its problem - p0
        const uint32_t bs = (uint32_t)b0 + lr;
        const bool carry = bs >= (uint32_t)B;
        prel = lq + (carry ? 1u : 0u);
        return (int32_t)(carry ? bs - (uint32_t)B : bs);
    };
    // stride % B == 0 (the host rounds the grid): this lane sees ONE budget -- its valid length is a constant of the launch and its
    // counters stay in registers
    const bool same_b = db == 0;
    const bool fixed_b = counters && same_b;
    uint32_t h1 = 0;
    unsigned long long tcs = 0;
    long long toks = 0;
    auto budget_len = [&](int32_t b) -> uint32_t { return nv_cached ? nv_lds[b] : (uint32_t)valid_len(a, b); };
    uint32_t prel0;
    const int32_t my_b = slot_budget(prel0);
    const uint32_t my_n = budget_len(my_b);

    uint32_t bad = 0;
    // The prefetch runs one step ahead of the counting with a scalar walker of its own (cf, pf, bf).
    int64_t cf = c0, pf = p0;
    int32_t bf = b0;
    // The truth of a step's cells travels with its images (a per-lane 4-byte LDS-DMA gather): the loop then holds no load the
    // compiler counts, so it cannot put a vmcnt(0) of its own in front of a use and drain the prefetch.
    auto issue_truth = [&]() {                                       // for the step at (cf, pf, bf); then the walker moves on
        const int64_t left = a.ncells - cf;
        const uint32_t live_cells = left > (int64_t)SC ? (uint32_t)SC : (uint32_t)left;
        // (scalar base + 32-bit lane offset; the walker is uniform but the compiler may keep it in VGPRs: an "s" operand it cannot
        //  satisfy is silently replaced by a VGPR pair, which does not assemble)
        const int32_t* tp = reinterpret_cast<const int32_t*>(sv_uniform64((int64_t)(uintptr_t)(a.truth + pf)));
        const uint32_t bs = (uint32_t)bf + lr;
        const uint32_t prel = lq + (bs >= (uint32_t)B ? 1u : 0u);
        sv_dma4(tp, ((uint32_t)lane < live_cells ? prel : 0u) * 4u, rbase + tru_off);
        cf += stride; pf += dp; bf += db;
        if (bf >= B) { bf -= B; pf += 1; }
    };
    auto advance = [&]() { c0 += stride; p0 += dp; b0 += db; if (b0 >= B) { b0 -= B; p0 += 1; } };
    int64_t st = wave;
    if (st < nsteps) { issue(st); issue_truth(); }
    // Stores a step issues AFTER the pieces of the next step's copy: the cell record and the cell's token sum (wave-uniform; one
    // instruction each -- a step has at least one live lane).  vmcnt retires in issue order on gfx9 (loads and stores share the counter;
    // hipcc's own waits rely on it), so "all but the last `late` operations" = every piece has landed, while the stores -- a full
    // write round trip, which a vmcnt(0) here exposed to the wave at every step -- stay in flight across the next step.
    const int late = (a.cells ? 1 : 0) + ((TOK && a.cell_tokens) ? 1 : 0);
    bool first = true;
#ifdef SCV_SORT_TIMELINE
    unsigned long long tl[6] = {0, 0, 0, 0, 0, 0}, tl_steps = 0;
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    for (; st < nsteps; st += nwaves) {
        SV_STAMP(5);                                                 // (loop control; the first stamp: everything before the loop)
        // this step's images have landed (LDS-DMA is counted by vmcnt; hipcc does not count asm loads)
        if (first || late == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (late == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        first = false;
        SV_STAMP(0);
        uint32_t R[NP];
        long long tok = 0;
        const int64_t left = a.ncells - c0;                          // (wave-uniform) cells from this step's first to the last
        const bool all_live = left >= (int64_t)SC;
        const uint32_t live_cells = all_live ? (uint32_t)SC : (uint32_t)left;
        const bool live = (uint32_t)lane < live_cells;
        uint32_t prel;
        const int32_t eb = same_b ? my_b : slot_budget(prel);
        const uint32_t n = live ? (same_b ? my_n : budget_len(eb)) : 0u;
        const int32_t trj = (int32_t)*reinterpret_cast<lds_u32*>((uintptr_t)(rbase + tru_off + (uint32_t)lane * 4u));
        const uint32_t ra = LIN ? rbase + shv + (uint32_t)lane * rowbytes : rbase + (uint32_t)lane * (PS * 16u);
        uint32_t w[NV];
        if (LIN) {
#pragma unroll
            for (int i = 0; i < NV; ++i) w[i] = *reinterpret_cast<lds_u32*>((uintptr_t)(ra + 4u * i));   // (past the row: the neighbour's votes, never valid)
        } else {
#pragma unroll
            for (int k = 0; k < RSM / 2; ++k) {
                // elements 4k .. 4k + 3 and their partners NP elements further on: both halves of four packed registers
                // (slots past the row re-read its last slot: in-domain values that are never valid votes)
                const uint32_t k0 = (uint32_t)k < RS ? (uint32_t)k : RS - 1u, k1 = (uint32_t)(k + RSM / 2) < RS ? (uint32_t)(k + RSM / 2) : RS - 1u;
                const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k0));
                const scv_v4u h = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + 16u * k1));
                w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
                w[NP + 4 * k] = h.x; w[NP + 4 * k + 1] = h.y; w[NP + 4 * k + 2] = h.z; w[NP + 4 * k + 3] = h.w;
            }
        }
        if (TOK && LIN) {
            const uint32_t rt = rbase + img_bytes + sht + (uint32_t)lane * rowbytes;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (i < N) {
                    const int32_t y = (int32_t)*reinterpret_cast<lds_u32*>((uintptr_t)(rt + 4u * i));
                    tok += (long long)(y & ((i - (int32_t)n) >> 31));      // validity as a mask: all ones when i < n
                }
            }
        }
        if (TOK && !LIN) {
#pragma unroll
            for (int k = 0; k < RSM; ++k) {
                if ((uint32_t)k < RS) {
                    const scv_v4u q = *reinterpret_cast<lds_v4u*>((uintptr_t)(ra + img_bytes + 16u * k));
                    // validity as a MASK (a predicate per element would keep 64 SGPR pairs alive): all ones when element < n
                    const int32_t nn = (int32_t)n - 4 * k;
                    tok += (long long)((int32_t)q.x & ((0 - nn) >> 31)) + (long long)((int32_t)q.y & ((1 - nn) >> 31))
                         + (long long)((int32_t)q.z & ((2 - nn) >> 31)) + (long long)((int32_t)q.w & ((3 - nn) >> 31));
                }
            }
        }
        // Two votes per register.  v_cvt_pk_u16_u32 saturates, so a slot outside 16 bits stays outside the domain and the domain check runs
        // on the packed registers (one v_or3 per four votes).  o1.py:140 int(extracted_answer) is unbounded; the extractor maps it into
        // bins 0..1023: the clamp runs only in the (wave-uniform, rare) case that some slot -- a vote or not -- holds a larger value
#pragma unroll
        for (int r = 0; r < NP; ++r) R[r] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(w[r], w[r + NP]));
        uint32_t orv = 0;
#pragma unroll
        for (int r = 0; r < NP; ++r) orv |= R[r];
        const bool full = all_live && __all(n == (uint32_t)NV);
        if (__any((orv & 0xfc00fc00u) != 0u)) {
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                // (validity as masks: all ones in a half whose slot index is below n)
                const uint32_t valid2 = ((uint32_t)((r - (int32_t)n) >> 31) & 0xffffu) | ((uint32_t)((r + NP - (int32_t)n) >> 31) << 16);
                bad |= R[r] & valid2;
                R[r] = pk_min_c(R[r], 0x03ff03ffu);
            }
        }
        // every slot is <= 1023 now (votes, neighbours, pads)
        if (!full) {
            const uint32_t n2 = n | (n << 16);
#pragma unroll
            for (int r = 0; r < NP; ++r)
                R[r] = sv_sentinel(R[r], n2, (uint32_t)(r + 1) | ((uint32_t)(r + NP + 1) << 16),
                                   (0x8000u | (uint32_t)r) | ((0x8000u | (uint32_t)(r + NP)) << 16));
        }
        SV_STAMP(1);
        // the next step's copy flies while this step is counted
        // (scalar bases of this step's outputs; a record is 16 bytes, or 4 under SCV_FLAG_PACKED_CELLS)
        char* const cells_out = a.cells ? reinterpret_cast<char*>(a.cells) + c0 * (a.packed_cells ? 4 : 16) : nullptr;
        int64_t* const ctok_out = (TOK && a.cell_tokens) ? a.cell_tokens + c0 : nullptr;
        advance();
        // The next step's copy goes into the image just read.  Votes only, 16 votes or more: its pieces are issued one every few
        // compare-exchanges of the sort -- pinned there through a register operand, or the compiler sinks the sort below them --
        // instead of back to back: 3-5 % (N = 64: 96.3 -> 92.1 us, N = 30: 86.2 -> 81.7).
        const bool have_next = st + nwaves < nsteps;
#ifdef SCV_SORT_NOSPREAD
        constexpr bool CAN_SPREAD = false;                           // (measurement build: the copy's pieces back to back, stamped on their own)
#else
        constexpr bool CAN_SPREAD = !TOK && NV >= 16;
#endif
        const bool spread = CAN_SPREAD && have_next;
        // (source base and limit of the next step's copy: wave-uniform values of this iteration)
        const int64_t nbyte0 = (st + nwaves) * SC * (int64_t)rowbytes;
        const int64_t nrem = LIN ? ((total_bytes - nbyte0 + shv - 1) & ~(int64_t)15) : total_bytes - nbyte0 - 16;
        const uint32_t nlim = nrem > 0x7fffffffll ? 0x7fffffffu : (nrem < 0 ? 0u : (uint32_t)nrem);
        const char* const ng = reinterpret_cast<const char*>(a.answers) + nbyte0 - shv;
        if (have_next) {
            if (spread) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue_truth(); }
            else { issue(st + nwaves); issue_truth(); }
        }
#ifdef SCV_SORT_NOSPREAD
        SV_STAMP(5);
#endif
        constexpr int STEP = sv_sort_ticks<NP>() / QMAX > 0 ? sv_sort_ticks<NP>() / QMAX : 1;
        int ticks = 0;                                               // (a constant at every call site after unrolling)
        auto piece = [&](int q, uint32_t& dep) __attribute__((always_inline)) {
            if constexpr (CAN_SPREAD) {
                // (readfirstlane: under register pressure the compiler may hold the uniform address in a VGPR, and an "s" operand it cannot
                //  satisfy is silently replaced by a VGPR, which does not assemble)
                if ((uint32_t)q < nq) sv_dma16_pinned(ng, off[q] < nlim ? off[q] : nlim, (uint32_t)__builtin_amdgcn_readfirstlane((int)(rbase + (uint32_t)q * 1024u)), dep);
            }
        };
        auto tick = [&](uint32_t& reg) __attribute__((always_inline)) {
            if constexpr (CAN_SPREAD) {
                if (spread && ticks % STEP == 0 && ticks / STEP < QMAX) piece(ticks / STEP, reg);
            }
            ++ticks;
        };
        if constexpr (CAN_SPREAD) {
            sv_sort<NP>(R, tick);
            if (spread) {
#pragma unroll
                for (int q = (sv_sort_ticks<NP>() + STEP - 1) / STEP; q < QMAX; ++q) piece(q, R[0]);   // (pieces the sort did not reach)
            }
        } else {
            SvNoTick none;
            sv_sort<NP>(R, none);
        }
        SV_STAMP(2);
        const uint32_t tcmp = (trj >= 0 && trj < kBins) ? (uint32_t)trj : 0x7fffu;
        const SortedStats s = sv_scan<NP>(R, tcmp | (tcmp << 16));
#ifdef SCV_SORT_TIMELINE
        asm volatile("" : : "v"(s.max_run), "v"(s.at_max), "v"(s.min_at_max), "v"(s.truth_votes));
#endif
        SV_STAMP(3);
        if (live) {
            const bool any = n > 0;
            const uint32_t maxc = any ? s.max_run : 0u;
            // a sentinel's key is above every vote's -- except in the 64-vote shape when every vote is distinct (run length 1 is key field 63
            // there: 63 << 10 | 1023 is the saturated key), where ALL sentinels were counted and come off again
            const uint32_t n_modes = any ? s.at_max - ((NV == 64 && s.max_run == 1u) ? (uint32_t)NV - n : 0u) : 0u;
            const uint32_t tc = s.truth_votes;
            const uint32_t hit = (any && tc == maxc) ? 1u : 0u;                  // o1.py:206
            if (a.cells) {
                uint4 rec;
                rec.x = maxc;
                rec.y = tc;
                rec.z = (n_modes & 0xffffu) | ((any ? (s.min_at_max & 0xffffu) : 0xffffu) << 16);
                rec.w = hit;
                if (a.packed_cells) __builtin_nontemporal_store(pack_cell(maxc, tc, n_modes, s.min_at_max, hit), reinterpret_cast<uint32_t*>(cells_out) + (uint32_t)lane);
                else __builtin_nontemporal_store(scv_v4u{rec.x, rec.y, rec.z, rec.w}, reinterpret_cast<scv_v4u*>(cells_out) + (uint32_t)lane);
            }
            if (TOK && a.cell_tokens) ctok_out[(uint32_t)lane] = tok;
            if (fixed_b) {                                                        // o1.py:238-240 as integers
                h1 += (hit && n_modes == 1u) ? 1u : 0u;
                if (hit && n_modes != 1u) atomicAdd(&tie[eb * TC + (int32_t)n_modes], 1u);
                tcs += tc;
                if (TOK) toks += tok;
            } else if (counters) {                                                // ... per workgroup in LDS
                if (hit) atomicAdd(&tie[eb * TC + (int32_t)n_modes], 1u);
                if (tc) atomicAdd(&acc[eb], (unsigned long long)tc);
                if (TOK) atomicAdd(&acc[B + eb], (unsigned long long)tok);
            }
        }
        SV_STAMP(4);
#ifdef SCV_SORT_TIMELINE
        ++tl_steps;
#endif
    }
#ifdef SCV_SORT_TIMELINE
    if (lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&scv_sort_timeline[i], tl[i]);
        atomicAdd(&scv_sort_timeline[6], tl_steps);
        atomicAdd(&scv_sort_timeline[7], 1ull);
    }
#endif
    if (bad & 0xfc00fc00u) atomicOr(a.err_flag, 1u);
    if (fixed_b) {
        if (h1) atomicAdd(&tie[my_b * TC + 1], h1);
        if (tcs) atomicAdd(&acc[my_b], tcs);
        if (TOK && toks) atomicAdd(&acc[B + my_b], (unsigned long long)toks);
    }
    if (counters) {
        __syncthreads();
        for (int64_t i = tid; i < (int64_t)B * TC; i += T) {
            const uint32_t v = tie[i];
            if (v && a.tie_hits) {
                const int64_t b = i / TC;
                atomicAdd(&a.tie_hits[b * SCV_TIE_CLASSES + (i - b * TC)], (unsigned long long)v);
            }
        }
        for (int i = tid; i < B; i += T) {
            if (a.truth_sum && acc[i]) atomicAdd(&a.truth_sum[i], acc[i]);
            if (TOK && a.token_sum && acc[B + i]) atomicAdd(&a.token_sum[i], acc[B + i]);
        }
    }
}

}  // namespace scv
