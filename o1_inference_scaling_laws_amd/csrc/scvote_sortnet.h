// scvote_sortnet.h -- the compare-exchange network of scv_sort_cells as a compile-time list.  Plain C++17 (no HIP): included by
// csrc/scvote_sort.hip.h for the device code and by tests/sortnet_check.cpp, which proves it on the CPU (0-1 principle).
#pragma once

namespace scv {

// Batcher's odd-even mergesort network on N wires, as a compile-time list of compare-exchanges (min to the lower wire): 5 / 19 / 63 /
// 191 of them for N = 4 / 8 / 16 / 32 against the bitonic network's 6 / 24 / 80 / 240.
template <int N>
struct SvNetwork {
    int a[N * 10], b[N * 10], n;
};
template <int N>
constexpr SvNetwork<N> sv_make_network() {
    SvNetwork<N> o{};
    o.n = 0;
    for (int p = 1; p < N; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= N - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) { o.a[o.n] = i + j; o.b[o.n] = i + j + k; ++o.n; }
    return o;
}

// Bitonic merge of N wires that hold a VALLEY (falling, then rising), ascending, as a compile-time list: the power-of-two network on
// the next power of two wires with +inf behind the N wires, minus every exchange that touches a pad (it never swaps) -- H. W. Lang's
// bitonic merge for arbitrary N.  52 exchanges for N = 24 (80 on 32 wires), 128 for N = 48.
constexpr int sv_floor_pow2_below(int n) { int m = 1; while (2 * m < n) m *= 2; return m; }     // greatest power of two < n (n >= 2)
template <int N>
constexpr SvNetwork<N> sv_make_valley_merge() {
    SvNetwork<N> o{};
    o.n = 0;
    int lo[2 * N] = {0}, len[2 * N] = {N}, top = 1;          // segments still to merge (a stack: the order of independent segments is free)
    while (top > 0) {
        --top;
        const int l = lo[top], n = len[top];
        if (n <= 1) continue;
        const int m = sv_floor_pow2_below(n);
        for (int i = l; i < l + n - m; ++i) { o.a[o.n] = i; o.b[o.n] = i + m; ++o.n; }
        lo[top] = l + m; len[top] = n - m; ++top;            // (popped after the lower segment)
        lo[top] = l; len[top] = m; ++top;
    }
    return o;
}
}  // namespace scv
