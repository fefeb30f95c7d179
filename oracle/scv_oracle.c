/*
 * scv_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's aggregation path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (o1_inference_scaling_laws_amd/) never does: it runs the HIP
 * kernels behind include/scvote.h or fails.
 *
 * What it restates (all paths relative to /root/reference):
 *   o1.py:181-195   vote collection: answers list of N ints, total_tokens = sum(tokens)
 *   o1.py:202       statistics.multimode(answers)  -- CPython stdlib, UNPINNED by the reference
 *                   (no requirements file); restated from /usr/lib/python3.10/statistics.py:586-601:
 *                   counts = Counter(data).most_common(); maxcount = counts[0][1];
 *                   return every value whose count == maxcount; [] for empty data.
 *   o1.py:204-213   score = 1/len(modes) if int(example['answer']) in modes else 0
 *   o1.py:229-245   accuracy = sum(score)/len(dataset); avg_tokens = mean(sum of tokens per problem)
 *   o1.py:274-276   budget b votes over the first n_valid[b] samples of its row
 *
 * Parity pinning: the reference has NO tests or golden vectors for this path and
 * its only inputs (helpers/response_cache.json) are a missing blob.  This restatement is pinned
 * instead against outputs of the UNMODIFIED reference run in the build container on synthetic
 * caches (tests/golden/make_golden.py -> tests/golden/golden_o1.json) and against
 * statistics.multimode itself (oracle/pyoracle.py), see tests/test_oracle_golden.py.
 * pass@k inputs (truth_count) and the bootstrap are NEW semantics: parity unpinned by the reference.
 *
 * The integer outputs are exactly those of include/scvote.h (same struct, same counters).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SCVO_BINS 1024
#define SCVO_TIE_CLASSES 1025

typedef struct scvo_cell {
    uint32_t max_count;
    uint32_t truth_count;
    uint16_t n_modes;
    int16_t  min_mode;
    uint8_t  hit;
    uint8_t  pad[3];
} scvo_cell;

/* ---- synthetic generator (spec in include/scvote.h, scv_synth_fill_i32) -------------------- */

static const uint64_t SCVO_G = 0x9E3779B97F4A7C15ull;

static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
static inline uint32_t mulhi32(uint32_t a, uint32_t n) { return (uint32_t)(((uint64_t)a * n) >> 32); }

typedef struct { uint32_t truth, q_num, d[4]; } scvo_pparam;

static scvo_pparam problem_params(uint64_t seed, int64_t p) {
    scvo_pparam r;
    uint64_t k = mix64((seed ^ 0x5851F42D4C957F2Dull) + SCVO_G * (uint64_t)(p + 1));
    r.truth = mulhi32((uint32_t)k, 1000u);
    r.q_num = 1u + (uint32_t)(k >> 32) % 7u;
    for (int j = 0; j < 4; ++j)
        r.d[j] = mulhi32((uint32_t)mix64(k + SCVO_G * (uint64_t)(j + 1)), 1000u);
    return r;
}

int scvo_synth_fill_i32(int32_t* answers, int32_t* tokens, int32_t* truth,
                        int64_t P, int32_t B, int64_t N, int64_t p_offset,
                        uint64_t seed, int dist) {
    if (P < 0 || B < 0 || N < 0 || dist < 0 || dist > 5) return -1;
    /* problems are independent (closed form per element): spread them over the host cores so that the
     * bench's wide parity check (hundreds of problems x 2^20 samples) does not wait on one core */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) if (P >= 4 && B * N >= (1 << 16))
#endif
    for (int64_t pl = 0; pl < P; ++pl) {
        const int64_t p = p_offset + pl;
        const scvo_pparam pp = problem_params(seed, p);
        if (truth) truth[pl] = (int32_t)pp.truth;
        if (!answers && !tokens) continue;
        const uint32_t t0 = pp.q_num * 429496729u;
        const uint32_t T5 = 214748364u;
        const uint32_t m = 2u + (uint32_t)(p & 1);
        const uint32_t base = ((p >> 1) & 1) ? (pp.truth + 500u) % 1000u : pp.truth;
        const int64_t full = (N / m) * m;
        for (int32_t b = 0; b < B; ++b) {
            const uint64_t e0 = ((uint64_t)p * (uint64_t)B + (uint64_t)b) * (uint64_t)N;
            int32_t* arow = answers ? answers + ((pl * B + b) * N) : NULL;
            int32_t* trow = tokens ? tokens + ((pl * B + b) * N) : NULL;
            for (int64_t i = 0; i < N; ++i) {
                const uint64_t u = mix64(seed + SCVO_G * (e0 + (uint64_t)i + 1));
                if (arow) {
                    const uint32_t hi = (uint32_t)(u >> 32), uv = mulhi32((uint32_t)u, 1000u);
                    uint32_t v;
                    switch (dist) {
                    case 0: v = uv; break;
                    case 1: {
                        if (hi < t0) v = pp.truth;
                        else { uint32_t x = hi - t0; v = (x < 4u * T5) ? pp.d[x / T5] : uv; }
                    } break;
                    case 2: v = pp.truth; break;
                    case 4: {   /* D4: D1 with a wrong value in the truth's place and the truth in d_0's */
                        const uint32_t hot = pp.d[0] == pp.truth ? (pp.truth + 500u) % 1000u : pp.d[0];
                        if (hi < t0) v = hot;
                        else { uint32_t x = hi - t0; uint32_t j = x / T5; v = (x < 4u * T5) ? (j == 0 ? pp.truth : pp.d[j]) : uv; }
                    } break;
                    case 5: v = (pp.truth + 500u) % 1000u; break;
                    default: v = (i < full) ? (base + 37u * (uint32_t)(i % m)) % 1000u
                                            : (base + 999u) % 1000u;
                    }
                    arow[i] = (int32_t)v;
                }
                if (trow) trow[i] = (int32_t)(100u + mulhi32((uint32_t)(mix64(u ^ SCVO_G) >> 32), 11901u));
            }
        }
    }
    return 0;
}

/* ---- the aggregation path ----------------------------------------------------------------- */

/*
 * Returns 0, or -2002 if a vote outside 0..1023 is seen and clamp == 0 (include/scvote.h
 * SCV_ERR_DOMAIN); with clamp != 0 such votes count for bin 1023 (SCV_FLAG_CLAMP_TO_INVALID_BIN).
 * Per-budget outputs are OVERWRITTEN.
 */
int scvo_aggregate_i32(const int32_t* answers, const int32_t* tokens,
                       const int32_t* n_valid, const int32_t* truth,
                       int64_t P, int32_t B, int64_t N, int clamp,
                       scvo_cell* cells_out, int64_t* cell_tokens_out,
                       int64_t* tie_class_hits_out, int64_t* token_sum_out,
                       int64_t* truth_count_sum_out) {
    if (P < 0 || B < 0 || N < 0 || (P * B > 0 && (!answers || !truth))) return -2001;
    if (tie_class_hits_out) memset(tie_class_hits_out, 0, sizeof(int64_t) * (size_t)B * SCVO_TIE_CLASSES);
    if (token_sum_out) memset(token_sum_out, 0, sizeof(int64_t) * (size_t)B);
    if (truth_count_sum_out) memset(truth_count_sum_out, 0, sizeof(int64_t) * (size_t)B);
    int bad = 0;
    uint32_t hist[SCVO_BINS];
    for (int64_t p = 0; p < P; ++p) {
        for (int32_t b = 0; b < B; ++b) {
            const int64_t cell = p * B + b;
            int64_t n = N;
            if (n_valid) { n = n_valid[b]; if (n < 0) n = 0; if (n > N) n = N; }
            const int32_t* row = answers + cell * N;
            /* o1.py:181-195: collect the votes (Counter of statistics.py:599) and sum the tokens */
            memset(hist, 0, sizeof hist);
            for (int64_t i = 0; i < n; ++i) {
                uint32_t v = (uint32_t)row[i];
                if (v > 1023u) { bad = 1; v = 1023u; }
                hist[v]++;
            }
            int64_t tok = 0;
            if (tokens) { const int32_t* tr = tokens + cell * N; for (int64_t i = 0; i < n; ++i) tok += tr[i]; }
            /* statistics.py:599-601: maxcount, then all values tied at maxcount */
            uint32_t maxc = 0;
            for (int v = 0; v < SCVO_BINS; ++v) if (hist[v] > maxc) maxc = hist[v];
            uint32_t n_modes = 0; int min_mode = -1;
            if (maxc > 0)
                for (int v = 0; v < SCVO_BINS; ++v)
                    if (hist[v] == maxc) { if (min_mode < 0) min_mode = v; n_modes++; }
            /* o1.py:206: int(example['answer']) in majority_answers */
            const int32_t t = truth[p];
            const uint32_t tc = (t >= 0 && t < SCVO_BINS) ? hist[t] : 0u;
            const uint8_t hit = (uint8_t)(maxc > 0 && tc == maxc);
            if (cells_out) {
                scvo_cell c; memset(&c, 0, sizeof c);
                c.max_count = maxc; c.truth_count = tc; c.n_modes = (uint16_t)n_modes;
                c.min_mode = (int16_t)min_mode; c.hit = hit;
                cells_out[cell] = c;
            }
            if (cell_tokens_out) cell_tokens_out[cell] = tok;
            /* o1.py:238-240: total_score += score; actual_tokens_used.append(tokens) -- kept as integers */
            if (tie_class_hits_out && hit) tie_class_hits_out[(int64_t)b * SCVO_TIE_CLASSES + n_modes] += 1;
            if (token_sum_out) token_sum_out[b] += tok;
            if (truth_count_sum_out) truth_count_sum_out[b] += tc;
        }
    }
    return (bad && !clamp) ? -2002 : 0;
}

/* Same computation with the problems spread over OpenMP threads (cells are independent, o1.py:234
 * already farms problems out to independent threads); used only for bench.py's all-cores baseline.
 * Per-budget outputs are reduced at the end, in problem order, so the result equals the 1-thread one. */
int scvo_aggregate_i32_mt(const int32_t* answers, const int32_t* tokens,
                          const int32_t* n_valid, const int32_t* truth,
                          int64_t P, int32_t B, int64_t N, int clamp, int threads,
                          scvo_cell* cells_out, int64_t* cell_tokens_out,
                          int64_t* tie_class_hits_out, int64_t* token_sum_out,
                          int64_t* truth_count_sum_out) {
    if (P < 0 || B < 0 || N < 0 || !cells_out || !cell_tokens_out) return -2001;
    int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) reduction(| : bad)
#endif
    for (int64_t p = 0; p < P; ++p) {
        const int rc = scvo_aggregate_i32(answers + p * B * N, tokens ? tokens + p * B * N : NULL, n_valid, truth + p, 1, B, N,
                                          clamp, cells_out + p * B, cell_tokens_out + p * B, NULL, NULL, NULL);
        if (rc == -2002) bad |= 1;          /* out-of-domain vote seen (cells still hold the clamped result) */
        else if (rc != 0) bad |= 2;
    }
    if (tie_class_hits_out) memset(tie_class_hits_out, 0, sizeof(int64_t) * (size_t)B * SCVO_TIE_CLASSES);
    if (token_sum_out) memset(token_sum_out, 0, sizeof(int64_t) * (size_t)B);
    if (truth_count_sum_out) memset(truth_count_sum_out, 0, sizeof(int64_t) * (size_t)B);
    for (int64_t p = 0; p < P; ++p)
        for (int32_t b = 0; b < B; ++b) {
            const scvo_cell* c = &cells_out[p * B + b];
            if (tie_class_hits_out && c->hit) tie_class_hits_out[(int64_t)b * SCVO_TIE_CLASSES + c->n_modes] += 1;
            if (token_sum_out) token_sum_out[b] += cell_tokens_out[p * B + b];
            if (truth_count_sum_out) truth_count_sum_out[b] += c->truth_count;
        }
    return (bad & 2) ? -2001 : ((bad & 1) ? -2002 : 0);
}

/* ---- problem-level bootstrap (spec in include/scvote.h, scv_bootstrap) ---------------------- */

int scvo_bootstrap(const scvo_cell* cells, int64_t P, int32_t B,
                   int32_t r_begin, int32_t r_end, uint64_t seed, int32_t M,
                   int64_t* counts_out) {
    if (P <= 0 || P > 0xFFFFFFFFll || B <= 0 || r_end < r_begin || M <= 0 || !cells || !counts_out) return -2001;
    int overflow = 0;
    memset(counts_out, 0, sizeof(int64_t) * (size_t)(r_end - r_begin) * (size_t)B * (size_t)M);
    for (int32_t r = r_begin; r < r_end; ++r) {
        int64_t* out = counts_out + (int64_t)(r - r_begin) * B * M;
        for (int64_t j = 0; j < P; ++j) {
            const uint64_t u = mix64(seed + SCVO_G * ((uint64_t)r * (uint64_t)P + (uint64_t)j + 1));
            const int64_t idx = (int64_t)mulhi32((uint32_t)(u >> 32), (uint32_t)P);
            for (int32_t b = 0; b < B; ++b) {
                const scvo_cell* c = &cells[idx * B + b];
                if (c->hit) { if (c->n_modes >= M) overflow = 1; else out[(int64_t)b * M + c->n_modes] += 1; }
            }
        }
    }
    return overflow ? -2001 : 0;
}
