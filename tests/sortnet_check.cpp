// CPU proof of the counting core of scv_sort_cells (csrc/scvote_sort.hip.h): built and run by tests/test_sort_network.py.
//  1. the compile-time compare-exchange list of csrc/scvote_sortnet.h sorts (0-1 principle: exhaustively for N <= 24 wires,
//     sampled for N = 32 here and EXHAUSTIVELY, bit-sliced, below: round 5), and the valley merge of the 48-vote shape sorts every 0-1 valley;
//  2. a scalar emulation of the device code's packed form -- two 16-bit elements per register, both halves through the same
//     network in lockstep, ONE bitonic merge whose first stage crosses the halves, then the run-start scan with the carry
//     between the halves, the keys (NV - run length) << 10 | value with their saturation, distinct sentinels behind the valid
//     prefix -- gives statistics.multimode's (max count, number of modes, smallest mode) and the truth count, against a
//     brute-force count, for every shape NV = 8 ... 64 (48: the halves meet at r = 0).
// The packed operations are restated here with the semantics of v_pk_min_u16 / v_pk_max_u16 / v_pk_add_u16 / v_pk_sub_u16 (clamp) /
// v_pk_mul_lo_u16 / v_pk_mad_u16 (clamp) / v_alignbit_b32 / v_bfi_b32; the order of operations is the device code's.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../o1_inference_scaling_laws_amd/csrc/scvote_sortnet.h"

using scv::sv_make_network;
using scv::sv_make_valley_merge;
using scv::SvNetwork;

template <typename F>
static uint32_t pk(uint32_t a, uint32_t b, F f) {
    return (uint32_t)(f(a & 0xffffu, b & 0xffffu) & 0xffffu) | ((uint32_t)(f(a >> 16, b >> 16) & 0xffffu) << 16);
}
static uint32_t pk_min(uint32_t a, uint32_t b) { return pk(a, b, [](uint32_t x, uint32_t y) { return x < y ? x : y; }); }
static uint32_t pk_max(uint32_t a, uint32_t b) { return pk(a, b, [](uint32_t x, uint32_t y) { return x > y ? x : y; }); }
static uint32_t pk_add(uint32_t a, uint32_t b) { return pk(a, b, [](uint32_t x, uint32_t y) { return x + y; }); }
static uint32_t pk_sub(uint32_t a, uint32_t b) { return pk(a, b, [](uint32_t x, uint32_t y) { return x - y; }); }
static uint32_t pk_sub_sat(uint32_t a, uint32_t b) { return pk(a, b, [](uint32_t x, uint32_t y) { return x > y ? x - y : 0u; }); }
static uint32_t pk_mul(uint32_t a, uint32_t b) { return pk(a, b, [](uint32_t x, uint32_t y) { return x * y; }); }
static uint32_t pk_mad_sat(uint32_t a, uint32_t b, uint32_t c) {     // v_pk_mad_u16 ... clamp
    auto one = [](uint32_t x, uint32_t y, uint32_t z) { const uint32_t v = x * y + z; return v > 0xffffu ? 0xffffu : v; };
    return one(a & 0xffffu, b & 0xffffu, c & 0xffffu) | (one(a >> 16, b >> 16, c >> 16) << 16);
}
static uint32_t alignbit(uint32_t hi, uint32_t lo, int s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> s); }

template <int N>
static bool network_sorts_all_01() {
    static constexpr SvNetwork<N> net = sv_make_network<N>();
    const uint64_t total = N <= 24 ? (1ull << N) : (1ull << 20);      // exhaustive up to 24 wires, SAMPLED (2^20 random 0-1 inputs) beyond
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (uint64_t t = 0; t < total; ++t) {
        uint64_t bits = t;
        if (N > 24) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; bits = rng; }
        int x[N];
        for (int i = 0; i < N; ++i) x[i] = (int)((bits >> i) & 1);
        for (int c = 0; c < net.n; ++c)
            if (x[net.a[c]] > x[net.b[c]]) { const int tmp = x[net.a[c]]; x[net.a[c]] = x[net.b[c]]; x[net.b[c]] = tmp; }
        for (int i = 1; i < N; ++i) if (x[i - 1] > x[i]) return false;
    }
    return true;
}

// the valley merge of csrc/scvote_sortnet.h (Lang's bitonic merge for arbitrary N: the pad exchanges left out) sorts EVERY 0-1 valley
// 1^a 0^b 1^c of N wires (0-1 principle restricted to the inputs the merge is applied to: falling, then rising) -- exhaustive
template <int N>
static bool valley_merge_sorts_all_01_valleys() {
    static constexpr SvNetwork<N> net = sv_make_valley_merge<N>();
    for (int a = 0; a <= N; ++a)
        for (int b = 0; a + b <= N; ++b) {
            int x[N];
            for (int i = 0; i < N; ++i) x[i] = (i < a || i >= a + b) ? 1 : 0;
            for (int c = 0; c < net.n; ++c)
                if (x[net.a[c]] > x[net.b[c]]) { const int tmp = x[net.a[c]]; x[net.a[c]] = x[net.b[c]]; x[net.b[c]] = tmp; }
            for (int i = 1; i < N; ++i) if (x[i - 1] > x[i]) return false;
        }
    return true;
}

template <int NV>
static bool packed_count_matches_bruteforce(int rounds) {
    constexpr int NP = NV / 2;
    constexpr bool MEET = (NP & (NP - 1)) != 0;                  // NP not a power of two: the halves meet at r = 0 (sv_sort, sv_scan)
    static constexpr SvNetwork<NP> net = sv_make_network<NP>();
    static constexpr SvNetwork<NP> valley = sv_make_valley_merge<NP>();
    uint64_t rng = 88172645463325252ull + NV;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int it = 0; it < rounds; ++it) {
        const int dom = (int[]){2, 3, 5, 50, 1024}[next() % 5];
        const bool corner = it % 16 == 7;                              // every vote distinct, 1023 among them: the key 63 << 10 | 1023 (NV = 64) is the saturated one
        const int pick = (int)(next() % 6);
        const uint32_t n = pick == 0 ? 0u : pick == 1 ? 1u : pick <= 3 ? (uint32_t)NV : (uint32_t)(next() % (NV + 1));
        uint32_t w[NV];
        for (int i = 0; i < NV; ++i) w[i] = corner ? (uint32_t)(1023 - i) : (uint32_t)(next() % dom);
        if (corner) for (int i = NV - 1; i > 0; --i) { const int j = (int)(next() % (uint64_t)(i + 1)); const uint32_t t = w[i]; w[i] = w[j]; w[j] = t; }
        const uint32_t truth = (next() & 1) ? w[next() % NV] : (uint32_t)(next() % 1024);
        // ---- device order: pack, sentinels, sort halves, cross merge, scan
        uint32_t R[NP];
        for (int r = 0; r < NP; ++r) R[r] = w[r] | (w[r + NP] << 16);
        if (n != (uint32_t)NV) {
            const uint32_t n2 = n | (n << 16);
            for (int r = 0; r < NP; ++r) {
                const uint32_t idx1 = (uint32_t)(r + 1) | ((uint32_t)(r + NP + 1) << 16), sent = (0x8000u | r) | ((0x8000u | (r + NP)) << 16);
                R[r] = pk_max(R[r], pk_mul(pk_min(pk_sub_sat(idx1, n2), 0x00010001u), sent));
            }
        }
        if (MEET) for (int r = 0; r < NP; ++r) R[r] ^= 0xffffu;                      // half 0 travels complemented
        for (int c = 0; c < net.n; ++c) { const uint32_t lo = R[net.a[c]], hi = R[net.b[c]]; R[net.a[c]] = pk_min(lo, hi); R[net.b[c]] = pk_max(lo, hi); }
        if (MEET) {
            for (int r = 0; r < NP; ++r) R[r] = pk_max(R[r], ~alignbit(R[r], R[r], 16));   // lo = ~min(a, b), hi = max(a, b)
            for (int c = 0; c < valley.n; ++c) { const uint32_t lo = R[valley.a[c]], hi = R[valley.b[c]]; R[valley.a[c]] = pk_min(lo, hi); R[valley.b[c]] = pk_max(lo, hi); }
            for (int r = 0; r < NP; ++r) R[r] ^= 0xffffu;
            // value order now: (0, NP-1) ... (0, 0), (1, 0) ... (1, NP-1)
            for (int r = 1; r < NP; ++r) if ((R[r - 1] & 0xffffu) < (R[r] & 0xffffu) || (R[r - 1] >> 16) > (R[r] >> 16)) { printf("NV=%d: halves not sorted\n", NV); return false; }
            if ((R[0] & 0xffffu) > (R[0] >> 16)) { printf("NV=%d: halves overlap\n", NV); return false; }
        } else {
            for (int r = 0; r < NP / 2; ++r) {
                uint32_t &a = R[r], &b = R[NP - 1 - r];
                const uint32_t t = alignbit(b, b, 16), mn = pk_min(a, t), mx = pk_max(a, t);
                a = (mn & 0xffffu) | (mx & 0xffff0000u);
                b = alignbit(mx, mn, 16);
            }
            for (int j = NP >> 1; j > 0; j >>= 1)
                for (int r = 0; r < NP; ++r) { const int l = r ^ j; if (l > r) { const uint32_t lo = R[r], hi = R[l]; R[r] = pk_min(lo, hi); R[l] = pk_max(lo, hi); } }
        }
        uint32_t run[NP], s = 0;
        for (int r = 0; r < NP; ++r) {
            const uint32_t prev = r ? R[r - 1] : ((R[MEET ? 0 : NP - 1] << 16) | 0xffffu);
            s = pk_max(s, pk_mul(pk_min(R[r] ^ prev, 0x00010001u), (uint32_t)(r + 1) | ((uint32_t)(r + NP + 1) << 16)));
            run[r] = s;
        }
        uint32_t carry = s << 16;
        if (MEET) {                                                                   // half 0's FIRST run continues into half 1
            uint32_t acc = 0;
            for (int r = 0; r < NP; ++r) acc = pk_add(acc, pk_min(run[r], 0x00020002u));
            carry = ((acc & 0xffffu) + 1u - (uint32_t)NP) << 16;
        }
        // key of an element = (NV - length of the run ending at it) << 10 | value: the smallest key is the longest run's last element with
        // the smallest value (max_count and min(multimode) in one packed minimum per register); a sentinel's key saturates (v_pk_mad_u16 clamp)
        uint32_t key[NP], kmin = 0xffffffffu;
        for (int r = 0; r < NP; ++r) {
            const uint32_t c2 = ((uint32_t)(NV - 1 - (r + 1)) & 0xffffu) | (((uint32_t)(NV - 1 - (r + NP + 1)) & 0xffffu) << 16);
            key[r] = pk_mad_sat(pk_add(pk_max(run[r], carry), c2), 0x04000400u, R[r]);
            kmin = pk_min(kmin, key[r]);
        }
        const uint32_t k1 = (kmin & 0xffffu) < (kmin >> 16) ? (kmin & 0xffffu) : (kmin >> 16);
        const uint32_t max_run = (uint32_t)NV - (k1 >> 10), thr = k1 | 0x3ffu, thr2 = thr | (thr << 16);
        const uint32_t tcmp = truth < 1024u ? truth : 0x7fffu, t2 = tcmp | (tcmp << 16);
        uint32_t above = 0, tc = 0;                                                   // (32-bit adds of packed 0 / 1: no carry between the halves)
        for (int r = 0; r < NP; r += 2) {
            above = above + pk_min(pk_sub_sat(key[r], thr2), 0x00010001u) + pk_min(pk_sub_sat(key[r + 1], thr2), 0x00010001u);
            tc = tc + pk_sub_sat(0x00010001u, R[r] ^ t2) + pk_sub_sat(0x00010001u, R[r + 1] ^ t2);
        }
        const uint32_t at_max = 2u * NP - ((above & 0xffffu) + (above >> 16));
        const bool any = n > 0;
        // (a sentinel is a run of one whose key saturates: above every vote's key except in the 64-vote shape when every vote is distinct --
        //  run length 1 is key field 63 there, 63 << 10 | 1023 = 0xffff -- where ALL sentinels count and are taken off again)
        const uint32_t got_max = any ? max_run : 0u, got_modes = any ? at_max - ((NV == 64 && max_run == 1u) ? (uint32_t)NV - n : 0u) : 0u;
        const uint32_t got_min = any ? (k1 & 0x3ffu) : 0xffffu;
        const uint32_t got_tc = (tc & 0xffffu) + (tc >> 16);
        // ---- brute force (statistics.multimode on the valid prefix)
        uint32_t cnt[1024] = {0}, want_max = 0, want_modes = 0, want_min = 0xffffu, want_tc = 0;
        for (uint32_t i = 0; i < n; ++i) { cnt[w[i]]++; want_tc += w[i] == truth; }
        for (int v = 0; v < 1024; ++v) if (cnt[v] > want_max) want_max = cnt[v];
        for (int v = 1023; v >= 0; --v) if (want_max && cnt[v] == want_max) { want_modes++; want_min = (uint32_t)v; }
        if (got_max != want_max || got_modes != want_modes || got_min != want_min || got_tc != want_tc) {
            printf("NV=%d n=%u: got (%u, %u, %u, %u) want (%u, %u, %u, %u)\n", NV, n, got_max, got_modes, got_min, got_tc, want_max, want_modes, want_min, want_tc);
            return false;
        }
    }
    return true;
}

// ---- round 5 ------------------------------------------------------------------------------------------------------------------------
// EXHAUSTIVE 0-1 proof of the 32-wire network, bit-sliced: a 64-bit word per wire carries 64 inputs at once (compare-exchange = and / or),
// 2^26 words cover all 2^32 inputs.  In the same pass the property scv_sort_prefix rests on: after merge phase p of the generator (p = 1, 2,
// 4 ... N / 2) EVERY aligned block of 2 p wires is sorted, for every input -- so the first 2 p wires hold the sorted first 2 p votes.
template <int N>
static int phase_end(int pmax) {                                        // exchanges of the phases p' <= pmax (the generator's own loops)
    int n = 0;
    for (int p = 1; p < N && p <= pmax; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= N - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) ++n;
    return n;
}
template <int N>
static bool network_sorts_all_01_bitsliced_with_phase_blocks() {
    static_assert(N >= 8 && N <= 32, "wires 0 .. 5 enumerate inside a word");
    static constexpr SvNetwork<N> net = sv_make_network<N>();
    int ends[8], nph = 0;
    for (int p = 1; p < N; p *= 2) ends[nph++] = phase_end<N>(p);
    if (ends[nph - 1] != net.n) return false;
    static const uint64_t low[6] = {0xAAAAAAAAAAAAAAAAull, 0xCCCCCCCCCCCCCCCCull, 0xF0F0F0F0F0F0F0F0ull, 0xFF00FF00FF00FF00ull, 0xFFFF0000FFFF0000ull, 0xFFFFFFFF00000000ull};
    const uint64_t words = 1ull << (N - 6);
    uint64_t badacc = 0;
    for (uint64_t t = 0; t < words; ++t) {
        uint64_t x[N];
        for (int i = 0; i < 6; ++i) x[i] = low[i];
        for (int i = 6; i < N; ++i) x[i] = ((t >> (i - 6)) & 1) ? ~0ull : 0ull;
        int c = 0;
        for (int ph = 0; ph < nph; ++ph) {
            for (; c < ends[ph]; ++c) { const uint64_t lo = x[net.a[c]] & x[net.b[c]], hi = x[net.a[c]] | x[net.b[c]]; x[net.a[c]] = lo; x[net.b[c]] = hi; }
            const int blk = 2 << ph;                                    // every aligned block of 2 p wires ascending: no 1 in front of a 0
            for (int i = 1; i < N; ++i) if (i % blk) badacc |= x[i - 1] & ~x[i];
        }
    }
    return badacc == 0;
}

// ... and the plain form for a wire count that is not a power of two (round 6: the 28-wire lockstep network of the 56-vote shape): all 2^N 0-1 inputs, bit-sliced
template <int N>
static bool network_sorts_all_01_bitsliced() {
    static_assert(N >= 8 && N <= 32, "wires 0 .. 5 enumerate inside a word");
    static constexpr SvNetwork<N> net = sv_make_network<N>();
    static const uint64_t low[6] = {0xAAAAAAAAAAAAAAAAull, 0xCCCCCCCCCCCCCCCCull, 0xF0F0F0F0F0F0F0F0ull, 0xFF00FF00FF00FF00ull, 0xFFFF0000FFFF0000ull, 0xFFFFFFFF00000000ull};
    const uint64_t words = 1ull << (N - 6);
    uint64_t badacc = 0;
    for (uint64_t t = 0; t < words; ++t) {
        uint64_t x[N];
        for (int i = 0; i < 6; ++i) x[i] = low[i];
        for (int i = 6; i < N; ++i) x[i] = ((t >> (i - 6)) & 1) ? ~0ull : 0ull;
        for (int c = 0; c < net.n; ++c) { const uint64_t lo = x[net.a[c]] & x[net.b[c]], hi = x[net.a[c]] | x[net.b[c]]; x[net.a[c]] = lo; x[net.b[c]] = hi; }
        for (int i = 1; i < N; ++i) badacc |= x[i - 1] & ~x[i];
    }
    return badacc == 0;
}

// scv_sort_prefix2: two sorted 64-element sequences A, B (element i = half i / 32 of register i % 32) -> flip stage across the files, then each
// file's bitonic merge (the stage between the halves of a register, five lockstep stages): ascending across A, then B.  0-1 principle on
// the inputs the merge is applied to (two sorted sequences: 65 x 65 pairs of zero counts), exhaustive; then the running 32-bit scan of the
// 128 sorted values and the key-free / keyed block scans against a brute-force statistics.multimode on random rows with sentinels.
static void flip_files(uint32_t* A, uint32_t* B, int NP) {
    for (int r = 0; r < NP; ++r) {
        const uint32_t t = alignbit(B[NP - 1 - r], B[NP - 1 - r], 16), mn = pk_min(A[r], t), mx = pk_max(A[r], t);
        A[r] = mn;
        B[NP - 1 - r] = alignbit(mx, mx, 16);
    }
}
static void merge_bitonic_file(uint32_t* X, int NP) {
    for (int r = 0; r < NP; ++r) {
        const uint32_t t = alignbit(X[r], X[r], 16), mn = pk_min(X[r], t), mx = pk_max(X[r], t);
        X[r] = (mn & 0xffffu) | (mx & 0xffff0000u);
    }
    for (int j = NP >> 1; j > 0; j >>= 1)
        for (int r = 0; r < NP; ++r) { const int l = r ^ j; if (l > r) { const uint32_t lo = X[r], hi = X[l]; X[r] = pk_min(lo, hi); X[l] = pk_max(lo, hi); } }
}
static uint32_t file_elem(const uint32_t* X, int NP, int i) { return i < NP ? (X[i] & 0xffffu) : (X[i - NP] >> 16); }
static bool two_file_merge_sorts_all_01_pairs() {
    constexpr int NP = 32;
    for (int za = 0; za <= 2 * NP; ++za)
        for (int zb = 0; zb <= 2 * NP; ++zb) {
            uint32_t A[NP] = {0}, B[NP] = {0};
            for (int i = 0; i < 2 * NP; ++i) {
                const uint32_t a = i < za ? 0u : 1u, b = i < zb ? 0u : 1u;
                A[i % NP] |= a << (16 * (i / NP)); B[i % NP] |= b << (16 * (i / NP));
            }
            flip_files(A, B, NP); merge_bitonic_file(A, NP); merge_bitonic_file(B, NP);
            uint32_t prev = 0;
            for (int i = 0; i < 4 * NP; ++i) { const uint32_t v = i < 2 * NP ? file_elem(A, NP, i) : file_elem(B, NP, i - 2 * NP); if (v < prev) return false; prev = v; }
        }
    return true;
}
struct Stats3 { uint32_t max_run, at_max, min_at_max; };
static Stats3 scan_running(const uint32_t* v, int n) {                  // sv_scan_files / sv_scan_block_running
    uint32_t prev = 0xffffffffu, len = 0, best = 0, cnt = 0, minv = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t x = v[i];
        len = x == prev ? len + 1u : 1u; prev = x;
        const bool gt = len > best;
        cnt = gt ? 1u : cnt + (len == best ? 1u : 0u);
        minv = gt ? x : minv;
        best = gt ? len : best;
    }
    return Stats3{best, cnt, minv};
}
static Stats3 scan_keyed(const uint32_t* v, int M) {                    // sv_scan_block
    uint32_t key[64], prev = 0xffffffffu, s = 0, km = 0xffffffffu;
    for (int i = 0; i < M; ++i) { const uint32_t x = v[i]; s = x != prev ? (uint32_t)i : s; key[i] = (((uint32_t)(M - 1 - i) + s) << 10) | x; km = key[i] < km ? key[i] : km; prev = x; }
    uint32_t cnt = 0;
    for (int i = 0; i < M; ++i) cnt += key[i] <= (km | 0x3ffu);
    return Stats3{(uint32_t)M - (km >> 10), cnt, km & 0x3ffu};
}
static bool prefix_scans_match_bruteforce(int rounds) {
    static constexpr SvNetwork<32> net = sv_make_network<32>();
    uint64_t rng = 0x1234567ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    auto brute = [](const uint32_t* w, int n, Stats3& o) {
        uint32_t cnt[1024] = {0}; o = Stats3{0, 0, 0xffffu};
        for (int i = 0; i < n; ++i) cnt[w[i]]++;
        for (int v = 0; v < 1024; ++v) if (cnt[v] > o.max_run) o.max_run = cnt[v];
        for (int v = 1023; v >= 0; --v) if (o.max_run && cnt[v] == o.max_run) { o.at_max++; o.min_at_max = (uint32_t)v; }
    };
    for (int it = 0; it < rounds; ++it) {
        const int dom = (int[]){1, 2, 3, 7, 50, 1024}[next() % 6];
        const int N = 68 + 4 * (int)(next() % 16);                      // 68 .. 128 votes
        uint32_t w[128];
        for (int i = 0; i < 128; ++i) w[i] = it % 9 == 4 ? (uint32_t)(1023 - i) : (uint32_t)(next() % dom);
        // half A: lockstep network with the block scans between the phases (keyed and running forms), then the halves merged
        uint32_t A[32], B[32];
        for (int r = 0; r < 32; ++r) A[r] = w[r] | (w[r + 32] << 16);
        int c = 0;
        for (int p = 1; p < 32; p *= 2) {
            for (const int e = phase_end<32>(p); c < e; ++c) { const uint32_t lo = A[net.a[c]], hi = A[net.b[c]]; A[net.a[c]] = pk_min(lo, hi); A[net.b[c]] = pk_max(lo, hi); }
            uint32_t blk[64]; Stats3 want;
            for (int i = 0; i < 2 * p; ++i) blk[i] = A[i] & 0xffffu;
            brute(w, 2 * p, want);
            const Stats3 g1 = scan_keyed(blk, 2 * p), g2 = scan_running(blk, 2 * p);
            if (g1.max_run != want.max_run || g1.at_max != want.at_max || g1.min_at_max != want.min_at_max ||
                g2.max_run != want.max_run || g2.at_max != want.at_max || g2.min_at_max != want.min_at_max) { printf("block scan of %d votes differs\n", 2 * p); return false; }
        }
        for (int r = 0; r < 16; ++r) { uint32_t &a = A[r], &b = A[31 - r]; const uint32_t t = alignbit(b, b, 16), mn = pk_min(a, t), mx = pk_max(a, t); a = (mn & 0xffffu) | (mx & 0xffff0000u); b = alignbit(mx, mn, 16); }
        for (int j = 16; j > 0; j >>= 1) for (int r = 0; r < 32; ++r) { const int l = r ^ j; if (l > r) { const uint32_t lo = A[r], hi = A[l]; A[r] = pk_min(lo, hi); A[l] = pk_max(lo, hi); } }
        // half B with distinct sentinels behind the row, sorted the same way
        const uint32_t nB = (uint32_t)N - 64u;
        for (int r = 0; r < 32; ++r) {
            B[r] = w[64 + r] | (w[96 + r] << 16);
            const uint32_t idx1 = (uint32_t)(r + 1) | ((uint32_t)(r + 33) << 16), sent = (0x8000u | r) | ((0x8000u | (r + 32)) << 16), n2 = nB | (nB << 16);
            if (nB != 64u) B[r] = pk_max(B[r], pk_mul(pk_min(pk_sub_sat(idx1, n2), 0x00010001u), sent));
        }
        for (int cc = 0; cc < net.n; ++cc) { const uint32_t lo = B[net.a[cc]], hi = B[net.b[cc]]; B[net.a[cc]] = pk_min(lo, hi); B[net.b[cc]] = pk_max(lo, hi); }
        for (int r = 0; r < 16; ++r) { uint32_t &a = B[r], &b = B[31 - r]; const uint32_t t = alignbit(b, b, 16), mn = pk_min(a, t), mx = pk_max(a, t); a = (mn & 0xffffu) | (mx & 0xffff0000u); b = alignbit(mx, mn, 16); }
        for (int j = 16; j > 0; j >>= 1) for (int r = 0; r < 32; ++r) { const int l = r ^ j; if (l > r) { const uint32_t lo = B[r], hi = B[l]; B[r] = pk_min(lo, hi); B[l] = pk_max(lo, hi); } }
        flip_files(A, B, 32); merge_bitonic_file(A, 32); merge_bitonic_file(B, 32);
        uint32_t all[128]; Stats3 want;
        for (int i = 0; i < 64; ++i) { all[i] = file_elem(A, 32, i); all[64 + i] = file_elem(B, 32, i); }
        for (int i = 1; i < 128; ++i) if (all[i - 1] > all[i]) { printf("two-file merge: not ascending\n"); return false; }
        Stats3 g = scan_running(all, 128);
        if (g.max_run == 1u) g.at_max -= 128u - (uint32_t)N;            // (sentinels are runs of one: they count only when every vote is distinct)
        brute(w, N, want);
        if (g.max_run != want.max_run || g.at_max != want.at_max || g.min_at_max != want.min_at_max) { printf("scan of %d votes: got (%u, %u, %u) want (%u, %u, %u)\n", N, g.max_run, g.at_max, g.min_at_max, want.max_run, want.at_max, want.min_at_max); return false; }
    }
    return true;
}

int main() {
    bool ok = true;
    ok &= network_sorts_all_01<2>() && network_sorts_all_01<4>() && network_sorts_all_01<8>() && network_sorts_all_01<16>();
    ok &= network_sorts_all_01<24>();                                   // the 48-vote shape's lockstep network: exhaustive (2^24 inputs)
    ok &= network_sorts_all_01<20>();                                   // the 40-vote shape's (round 6): exhaustive (2^20 inputs)
    ok &= network_sorts_all_01<12>();                                   // the 24-vote shape's (round 6): exhaustive
    ok &= network_sorts_all_01_bitsliced<28>();                         // the 56-vote shape's (round 6): exhaustive, bit-sliced (2^28 inputs)
    ok &= network_sorts_all_01<32>();                                   // (sampled: 2^20 random 0-1 inputs; all 2^32 below, bit-sliced)
    printf("network: %s (exchanges on 4 / 8 / 16 / 24 / 32 wires: %d %d %d %d %d; exhaustive up to 24 wires, 32 sampled here and exhaustive below)\n", ok ? "sorts" : "FAILS", sv_make_network<4>().n,
           sv_make_network<8>().n, sv_make_network<16>().n, sv_make_network<24>().n, sv_make_network<32>().n);
    const bool okv = valley_merge_sorts_all_01_valleys<6>() && valley_merge_sorts_all_01_valleys<12>() && valley_merge_sorts_all_01_valleys<24>() &&
                     valley_merge_sorts_all_01_valleys<16>() && valley_merge_sorts_all_01_valleys<48>() && valley_merge_sorts_all_01_valleys<20>() &&
                     valley_merge_sorts_all_01_valleys<40>() && valley_merge_sorts_all_01_valleys<28>() && valley_merge_sorts_all_01_valleys<56>();
    printf("valley merge: %s (exchanges on 24 wires: %d)\n", okv ? "sorts every 0-1 valley" : "FAILS", sv_make_valley_merge<24>().n);
    ok &= okv;
    bool ok2 = packed_count_matches_bruteforce<8>(20000) && packed_count_matches_bruteforce<16>(20000) && packed_count_matches_bruteforce<32>(20000) &&
               packed_count_matches_bruteforce<48>(60000) && packed_count_matches_bruteforce<64>(20000) && packed_count_matches_bruteforce<24>(20000) &&
               packed_count_matches_bruteforce<40>(60000) && packed_count_matches_bruteforce<56>(60000);
    printf("packed sort + scan: %s\n", ok2 ? "equals statistics.multimode" : "DIFFERS");
    const bool ok3 = network_sorts_all_01_bitsliced_with_phase_blocks<32>() && network_sorts_all_01_bitsliced_with_phase_blocks<16>();
    printf("32 wires, all 2^32 inputs, every aligned block of 2 p wires after phase p: %s\n", ok3 ? "sorted" : "FAILS");
    const bool ok4 = two_file_merge_sorts_all_01_pairs();
    printf("two-file merge: %s\n", ok4 ? "sorts every pair of sorted 0-1 halves" : "FAILS");
    const bool ok5 = prefix_scans_match_bruteforce(20000);
    printf("prefix block scans + 128-vote scan: %s\n", ok5 ? "equal statistics.multimode" : "DIFFER");
    return ok && ok2 && ok3 && ok4 && ok5 ? 0 : 1;
}
