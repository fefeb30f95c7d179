/*
 * scvote.h -- C ABI of the MI355X-native self-consistency aggregation engine.
 *
 * This is the drop-in boundary for ONE hot path of hughbzhang/o1_inference_scaling_laws:
 * the majority-vote / tie-aware scoring / per-budget reduction behind
 *
 *     o1.py:167  process_single_example(example, token_limit, cache, N) -> (score, total_tokens)
 *     o1.py:216  run_experiments(dataset, cache, token_limit, N)        -> (accuracy, avg_tokens_used)
 *
 * The reference has no FFI of its own (it is 421 lines of Python); these entry points are what a
 * ctypes binding for that path binds (see INTEGRATION.md for the stub a maintainer adds to o1.py).
 * Everything here is plain C: pointers, sizes, fixed-width integers.  No torch / numpy types.
 *
 * Data model:
 *   answers  int32 [P, B, N]  row-major, N contiguous.  One element = one sample-vote.
 *                             Value domain: bins 0..1023 (AIME answers 0..999 + 24 spare bins the
 *                             host-side extractor uses to dictionary-encode out-of-domain ints).
 *   tokens   int32 [P, B, N]  optional second stream (completion tokens of each sample).
 *   n_valid  int32 [B]        optional: budget b votes over the prefix answers[p,b,0:n_valid[b]]
 *                             (o1.py:274-276: N = token_limit // min(2**11, token_limit)).
 *   truth    int32 [P]        ground-truth bin of each problem (o1.py:206 int(example['answer'])).
 *
 * Per (problem, budget) cell the engine produces one scv_cell; per budget it produces integer
 * counters from which the host computes the reference's floats in ONE canonical order
 * (o1.py:244-245).  All device outputs are integers => bit-exact at any GPU count.
 */
#ifndef SCVOTE_H
#define SCVOTE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCV_NUM_BINS 1024         /* histogram bins per cell (0..999 answers + 24 spare)        */
#define SCV_TIE_CLASSES 1025      /* tie_class_hits[b][m], m = n_modes in 0..1024               */

/* mem_kind: where every pointer argument of a call lives. */
#define SCV_MEM_HOST 0            /* host pointers; staged through HBM in problem-chunks by a 3-stage
                                     pipeline (threads -> pinned bounce -> DMA -> kernel); blocking */
#define SCV_MEM_DEVICE 1          /* device pointers; async on the ctx stream, caller syncs     */

/* scv_create flags */
#define SCV_FLAG_TIMING 0x1u      /* record hipEvents around every aggregation kernel launch    */
#define SCV_FLAG_CLAMP_TO_INVALID_BIN 0x2u /* out-of-domain votes go to bin 1023, no error      */
#define SCV_FLAG_PACKED_CELLS 0x4u /* OPT-IN: scv_aggregate_i32 (DEVICE memory, N <= 127) writes its cell table as uint32_t [P, B], 4 bytes per cell:
                                      max_count | truth_count << 7 | n_modes << 14 | min_mode << 21 | hit << 31 (7 + 7 + 7 + 10 + 1 bits; an empty cell
                                      -- max_count 0 -- has min_mode field 1023 and means -1).  The reference's most common call is N = 1
                                      (o1.py:302) and N = 1, 2, 4, 8 (o1.py:276): a 16-byte scv_cell per 4-byte vote is 80 % of such a launch's traffic.
                                      Everything else (counters, cell_tokens) is unchanged; a call the flag does not cover (prefix budgets, HOST memory,
                                      N > 127, a forced streaming path, vote + bootstrap) with cells_out != NULL is SCV_ERR_ARG.  Decoder:
                                      o1_inference_scaling_laws_amd/engine.py unpack_cells */

/* synthetic distributions */
#define SCV_DIST_UNIFORM 0        /* D0: uniform over 0..999                                    */
#define SCV_DIST_PEAKED 1         /* D1: truth w.p. q_p in {.1...7}, 4 distractors at .05, rest uniform */
#define SCV_DIST_DEGENERATE 2     /* D2: every vote == truth[p]                                 */
#define SCV_DIST_TIE 3            /* D3: exact 2-/3-way ties, with and without truth among them */
#define SCV_DIST_PEAKED_WRONG 4   /* D4: D1 with the roles swapped: a WRONG value w.p. q_p, truth at .05 -- a confidently
                                     wrong majority, an ordinary outcome of the reference (o1.py:204-213 scores it 0)    */
#define SCV_DIST_DEGENERATE_WRONG 5 /* D5: every vote == one wrong value (truth + 500) % 1000                            */

/* error codes: 0 ok; -(hipError_t) for HIP failures; the following for argument/data errors. */
#define SCV_OK 0
#define SCV_ERR_ARG (-2001)           /* null/negative/oversized argument                       */
#define SCV_ERR_DOMAIN (-2002)        /* a vote outside 0..1023 was seen (results are invalid)  */
#define SCV_ERR_NO_DEVICE (-2003)     /* no HIP device visible                                  */
#define SCV_ERR_NOT_TIMED (-2004)     /* scv_last_kernel_ns without SCV_FLAG_TIMING / no launch */
#define SCV_ERR_ALLOC (-2005)         /* device allocation failed                               */

/*
 * Result of one (problem, budget) cell.  Replaces the return value of
 * statistics.multimode(answers) + the scoring at o1.py:202-213:
 *   score = hit ? 1.0 / n_modes : 0          (o1.py:206,210)
 * truth_count (= histogram[truth]) is the c of the pass@k estimator.
 */
typedef struct scv_cell {
    uint32_t max_count;   /* count of the modal value(s); 0 iff the cell had no valid votes   */
    uint32_t truth_count; /* votes equal to truth[p]                                           */
    uint16_t n_modes;     /* len(statistics.multimode(answers)); 0 iff max_count == 0          */
    int16_t  min_mode;    /* smallest modal bin, -1 iff max_count == 0                         */
    uint8_t  hit;         /* truth[p] in multimode(answers)                                    */
    uint8_t  pad[3];
} scv_cell;               /* 16 bytes */

typedef struct scv_ctx scv_ctx;   /* opaque: device, stream, scratch, timing events           */

/* Create a context bound to `device` (-1 = current HIP device).  One ctx per GPU per process. */
int scv_create(scv_ctx** out, int device, uint32_t flags);
int scv_destroy(scv_ctx* ctx);

/* A new ctx launches on a private non-blocking stream.  scv_set_stream BORROWS the caller's
 * hipStream_t instead (e.g. torch.cuda.current_stream().cuda_stream); NULL is the device's default
 * stream -- which is what torch hands out until the user opens a stream of their own.  Does not
 * synchronise (legal while the new stream is being captured into a hipGraph): every DEVICE-mode
 * entry point only enqueues work, so aggregation + bootstrap can be captured once and replayed. */
int scv_set_stream(scv_ctx* ctx, void* hip_stream);
/* Block until everything queued on the ctx stream has finished; reports SCV_ERR_DOMAIN if a
 * DEVICE-mode aggregation since the last sync saw an out-of-domain vote. */
int scv_sync(scv_ctx* ctx);

/*
 * Streaming-kernel geometry (for A/B measurement; 0 / negative = keep current).  By default the library picks the geometry
 * from the shape (measured bands); any explicit value here switches that off until ALL FOUR arguments are negative
 * (scv_set_tuning(ctx, -1, -1, -1, -1): back to the library's own choice).  Instantiated: (copies, threads) in (4, 256) with unroll 2; (8, 256) (8, 512)
 * (16, 256) (16, 512) (16, 1024) with unroll 4 -- anything else is SCV_ERR_ARG.
 *   copies        LDS sub-histogram replication R        threads     workgroup size
 *   wg_per_cu     persistent workgroups per CU (clamped by LDS and wave capacity)
 *   unroll        16-byte loads in flight per lane
 */
int scv_set_tuning(scv_ctx* ctx, int copies, int threads, int wg_per_cu, int unroll);
/*
 * Launch options.  None is needed for correct results: the defaults are the measured choices (DESIGN_HISTORY.md 4).  The keys exist
 * so that every kernel of the family can be forced (parity tests) and the dispatch bounds re-measured; every choice that was
 * measured slower in rounds 1-3 is gone from the library (DESIGN_HISTORY.md 4 lists them with their numbers).
 *   "overwrite_counters"  default 0; 1: DEVICE-mode per-budget counters are OVERWRITTEN, not accumulated into (with few long cells the
 *                         streaming kernel's last workgroup does it and the call is one launch, otherwise a memset precedes the launch)
 *   "path"                0 auto | 1 streaming, whole cells | 2 streaming, split-N + merge | 4 register-resident cells | 5 sorted cells
 *   "sort_n_min" / "sort_n_max"   defaults 8 / 64: cells of that many votes run one lane per cell, the wave's rows staged through LDS
 *                         by LDS-DMA and sorted in registers (rows that are not 16-byte aligned: from 5); "sort_n_max" = 0: off
 *   "reg_n_max"           default 8192 = its maximum: 32 < N <= this uses the register-resident cell kernels; 0: the streaming kernel
 *   "reg_shape"           force a register-resident shape: g * 100 + v (g lanes per cell, v vectors per lane: 1601 803 804 1602 1604 3204 6404;
 *                         803 / 804 = 8 lanes x 3 / 4 vectors with 8-bit LDS bins, the auto choice for 65 .. 96 / 97 .. 128 slots since round 6 --
 *                         1602 there is round 5's 128-slot shape) or 1041 / 1042 / 1044 / 1048 (dense scan, 1, 2, 4, 8 parts of 4 KiB); a shape too
 *                         small for N is ignored
 *   "auto_geometry"       DEPRECATED alias kept for callers of rounds 1-4: 1 = scv_set_tuning(ctx, -1, -1, -1, -1) (the library picks the
 *                         streaming geometry from the shape), 0 = pin the current geometry
 *   "fused_counters_max"  default 512 (4096 until round 6; re-measured by tools/crossovers.py): streaming kernel -- at or below this many cells (or problem rows >= 4 MiB) per-cell atomics inside
 *                         the hot kernel, otherwise a separate reduction of the cell table; 0: EVERY kernel leaves the counters to that
 *                         reduction (the cell kernels otherwise keep per-workgroup LDS tables and flush them in the same launch)
 *   "grid"                > 0: exact persistent grid (0: from the CU count, balanced so that all workgroups stream the same number of items)
 *   "segs"                split-N segments per cell (0 auto)
 *   "prefix_path"         scv_aggregate_prefix_i32: 0 auto | 1 one lane per problem, every budget out of one pass (pools <= 64) |
 *                         2 the cell kernels on pool rows (pools <= 4096) | 3 one streaming pass, a histogram snapshot per boundary |
 *                         4 one pass per problem over its pool row, 16 / 32 / 64 lanes per problem, every budget a snapshot of the running
 *                         mode statistics (pools <= 4096: auto above 64 votes and for budget lists too long for path 1; "reg_shape" = 16 / 32 forces the lanes per problem)
 *                         auto includes: pools of 17 .. 128 votes (N % 4 == 0, 16-byte aligned bases) whose budgets are all 0, a power of two <= 16 / 32 / 64
 *                         (pools <= 32 / 64 / 128), or >= N come out of ONE sort per problem (scv_sort_prefix) -- pools of 17 .. 64 votes always, pools of
 *                         68 .. 128 votes (scv_sort_prefix2: ~27 us for a launch of one step per wave) from 57 344 pools per call; with tokens their sums
 *                         come from token steps of the same launch (scv_sort_prefix2<true>).  A HOST-mode call reads the budgets; a DEVICE-mode call queues that
 *                         kernel in front of the general one and the two decide from n_valid which of them does the work (~4 us for the one that
 *                         leaves).  | 5 = auto, and the caller PROMISES budgets of that form for such pools (any number of them, with or without
 *                         tokens): a DEVICE-mode call queues scv_sort_prefix alone; a list that breaks the promise computes nothing and is
 *                         SCV_ERR_ARG (at the call in HOST mode, at the next synchronisation in DEVICE mode).  1 .. 4 switch the kernel off
 *   "boot_path"           scv_aggregate_bootstrap_i32: 0 auto (ONE cooperative launch when the shape allows it) | 1 one ORDINARY launch |
 *                         2 two launches, LDS-resident code table | 3 two launches, global gathers (also scv_bootstrap's kernel)
 *   "boot_spin_limit"     default 2^20: polls at the grid barrier before a workgroup of an ordinary one-launch form gives up and leaves
 *                         the bootstrap to scv_sync; tests set 1
 *   "stage_mb"            HOST mode: chunk size of the staging pipeline (default 128)
 *   "host_small_kb"       HOST mode: calls whose inputs + outputs fit in this many KiB (default 1024 -- every call the reference itself
 *                         makes: P = 30, N <= 128, o1.py:277,302) skip the pipeline: one pinned block, one H2D, the kernel, one D2H, one
 *                         stream sync, no threads and no per-call allocation; 0: every HOST call goes through the pipeline
 *   "copy_threads"        HOST mode: threads filling the pinned bounce slots (default 6)
 */
int scv_set_option(scv_ctx* ctx, const char* key, int64_t value);

/*
 * The hot path.  Replaces, for all P problems and B budgets at once,
 *   o1.py:181-195 (vote collection + token sum), o1.py:202 (statistics.multimode),
 *   o1.py:204-213 (tie-aware score) and the integer part of o1.py:229-245 (per-budget sums).
 *
 * Outputs (any may be NULL):
 *   cells_out            scv_cell [P, B]
 *   cell_tokens_out      int64    [P, B]   sum of tokens over the valid prefix (o1.py:195)
 *   tie_class_hits_out   int64    [B, 1025] #problems with hit and n_modes == m  (o1.py:238-239)
 *   token_sum_out        int64    [B]       sum over problems of cell_tokens     (o1.py:240,245)
 *   truth_count_sum_out  int64    [B]       sum over problems of truth_count
 * In DEVICE mode the three per-budget outputs are ACCUMULATED INTO (+=), so a caller streaming
 * problem-chunks (or ranks before an all-reduce) zeroes them once; in HOST mode they are overwritten.
 */
int scv_aggregate_i32(scv_ctx* ctx,
                      const int32_t* answers, const int32_t* tokens,
                      const int32_t* n_valid, const int32_t* truth,
                      int64_t P, int32_t B, int64_t N, int mem_kind,
                      scv_cell* cells_out, int64_t* cell_tokens_out,
                      int64_t* tie_class_hits_out, int64_t* token_sum_out,
                      int64_t* truth_count_sum_out);

/*
 * Prefix budgets over ONE sample pool per problem.  The reference's budgets
 * T >= 2^11 vote over prefixes of the same samples (o1.py:274-277 with the idx-keyed cache of
 * o1.py:85-88).  pool / tokens are int32 [P, N]; cell (p, b) votes over pool[p, 0:n_valid[b]]
 * (n_valid required, any order, B <= 512).  Outputs and conventions are exactly those of
 * scv_aggregate_i32 called on the dense expansion answers[p, b, :] = pool[p, :], but the pool is
 * read from HBM once: 4 * max_b n_valid[b] algorithmic bytes per problem instead of 4 * sum_b n_valid[b].
 * Dispatch by pool length: N <= 64 (the reference's pools) one lane per problem, every budget a snapshot of one
 * pass over its votes; N <= 4096 the cell kernels, each cell reading its prefix of the problem's row; longer pools
 * the streaming kernel with a snapshot of the histogram at every boundary (option "prefix_path" forces one).
 */
int scv_aggregate_prefix_i32(scv_ctx* ctx,
                             const int32_t* pool, const int32_t* tokens,
                             const int32_t* n_valid, const int32_t* truth,
                             int64_t P, int32_t B, int64_t N, int mem_kind,
                             scv_cell* cells_out, int64_t* cell_tokens_out,
                             int64_t* tie_class_hits_out, int64_t* token_sum_out,
                             int64_t* truth_count_sum_out);

/*
 * Problem-level bootstrap (new semantics, not in the reference).  For resample
 * r in [r_begin, r_end): draw P problem indices idx_j = mulhi32(hi32(mix64(seed + G*(r*P+j+1))), P)
 * and count, per budget, hits by tie class.  counts_out int64 [r_end-r_begin, B, M]; a drawn hit
 * with n_modes >= M returns SCV_ERR_ARG (pick M = 1 + largest class present in tie_class_hits).
 * cells is scv_cell [P, B].  Pointers follow mem_kind.
 */
int scv_bootstrap(scv_ctx* ctx, const scv_cell* cells, int64_t P, int32_t B,
                  int32_t r_begin, int32_t r_end, uint64_t seed, int32_t M, int mem_kind,
                  int64_t* counts_out);

/*
 * Vote + bootstrap in ONE call (DEVICE pointers; BASELINE config 5 on one GPU): scv_aggregate_i32 followed by
 * scv_bootstrap over the cell table it has just written, with the same outputs as the two calls.  When the shape
 * allows it (whole-cell streaming kernel, the [P, B] code table fits the workgroup's LDS, the persistent grid is
 * resident at once) both run in ONE kernel launch: all workgroups meet at a grid barrier after their last cell and
 * then share the resamples.  Otherwise (or with option "boot_path" >= 2) the bootstrap kernel is queued behind the
 * vote kernel on the same stream.  cells_out and counts_out are required.  Asynchronous like every DEVICE-mode call.
 * Co-tenancy cannot make a valid call fail: the one-launch form is started with hipLaunchCooperativeKernel (the
 * runtime starts it only when the whole grid is resident, whatever other streams, RCCL or other processes run; a
 * refused cooperative launch becomes two launches; option "boot_path" = 1 uses an ordinary launch, as does a
 * call made while the stream is being captured into a hipGraph).  Under an ordinary launch a workgroup gives up at the
 * barrier after "boot_spin_limit" polls; the vote outputs are complete by then, and the next scv_sync resets the
 * barrier, runs scv_bootstrap over cells_out as a separate launch and returns its status (stat "boot_recovered").
 * Until that scv_sync counts_out is undefined -- as after any asynchronous call that has not been synchronised.
 */
int scv_aggregate_bootstrap_i32(scv_ctx* ctx,
                                const int32_t* answers, const int32_t* tokens,
                                const int32_t* n_valid, const int32_t* truth,
                                int64_t P, int32_t B, int64_t N,
                                scv_cell* cells_out, int64_t* cell_tokens_out,
                                int64_t* tie_class_hits_out, int64_t* token_sum_out,
                                int64_t* truth_count_sum_out,
                                int32_t r_begin, int32_t r_end, uint64_t seed, int32_t M,
                                int64_t* counts_out);

/*
 * Closed-form integer synthetic generator, evaluated ON DEVICE (config C3 is 335.5 GB and cannot
 * be shipped).  Element (p, b, i) depends only on (seed, dist, p_offset + p, b, i, B, N), so shards
 * and the CPU mirror (o1_inference_scaling_laws_amd/synth.py) produce identical tensors.
 *   mix64 = splitmix64 finaliser, G = 0x9E3779B97F4A7C15, mulhi32(a, n) = (a * n) >> 32
 *   k_p   = mix64((seed ^ 0x5851F42D4C957F2D) + G * (p + 1))
 *   truth = mulhi32(lo32(k_p), 1000);  q_num = 1 + hi32(k_p) % 7;  d_j = mulhi32(lo32(mix64(k_p + G*(j+1))), 1000)
 *   u     = mix64(seed + G * (((p*B + b) * N + i) + 1)),  uv = mulhi32(lo32(u), 1000)
 *   D0: uv.  D1: x = hi32(u); x < q_num*429496729 -> truth; else (x - that) < 4*214748364 -> d_[(x-that)/214748364]; else uv.
 *   D2: truth.  D3: m = 2 + (p & 1); base = ((p >> 1) & 1) ? (truth + 500) % 1000 : truth;
 *       i < (N / m) * m -> (base + 37 * (i % m)) % 1000, else (base + 999) % 1000.
 *   D4: D1 with hot = (d_0 == truth ? (truth + 500) % 1000 : d_0) in the truth's place and truth in d_0's:
 *       x < q_num*429496729 -> hot; else j = (x-that)/214748364 < 4 -> (j == 0 ? truth : d_j); else uv.
 *   D5: (truth + 500) % 1000.
 *   tokens: 100 + mulhi32(hi32(mix64(u ^ G)), 11901)                       (100..12000)
 * All pointers are DEVICE pointers (any may be NULL); async on the ctx stream.
 */
int scv_synth_fill_i32(scv_ctx* ctx, int32_t* answers, int32_t* tokens, int32_t* truth,
                       int64_t P, int32_t B, int64_t N, int64_t p_offset,
                       uint64_t seed, int dist);

/*
 * The device error word (bit 0: a vote outside bins 0..1023 -- exported as 0 under SCV_FLAG_CLAMP_TO_INVALID_BIN, where
 * scv_sync does not treat it as an error either; bit 1: a drawn bootstrap hit had n_modes >= M; bit 2: the one-launch vote +
 * bootstrap of a NON-cooperative launch gave up at its grid barrier -- not an error: the next scv_sync repairs it), widened to
 * int64 and written to *dst_device IN STREAM ORDER behind everything queued so far; it is not cleared (scv_sync does
 * that).  For multi-GPU callers: the reference sums scores over problems in one process (o1.py:236-245); when the
 * problems are sharded over ranks, the word goes into one extra element behind the packed counters so that the ONE
 * all-reduce of the evaluation also tells every rank whether any rank's counters are invalid -- no host round trip,
 * no rank left waiting in a collective (o1_inference_scaling_laws_amd/passk.py, dist.py).
 */
int scv_export_error_word(scv_ctx* ctx, int64_t* dst_device);

/*
 * Several GPUs from ONE process (the reference is one process, o1.py:312-315): a communicator owns one scv_ctx per device and
 * the exchange step of the path -- the sum over problems of o1.py:236-245, which becomes ONE all-reduce of the packed int64
 * per-budget counters when the problems are sharded by contiguous block over the devices.  No launcher, no
 * torch.distributed: a ctypes / C caller shards its problems, calls scv_aggregate_i32 (DEVICE mode) on every rank's ctx, then
 * scv_allreduce_counters, then scv_comm_sync.
 *
 *   devices      HIP device index of every rank (NULL or n == 0: all visible devices, rank r = device r); an index may
 *                repeat (several contexts on one GPU: how a 1-GPU box exercises the path); at most 16 ranks
 *   ctx_flags    SCV_FLAG_* for every rank's context
 *   comm_flags   SCV_COMM_PEER (default): one-shot all-reduce over xGMI peer access, pure HIP -- every rank's kernel reads the
 *                other ranks' buffers directly, sums them into a staging buffer, and the sums replace the buffers when every
 *                rank has finished reading; the devices' streams are ordered by events, the host never blocks.
 *                SCV_COMM_RCCL: ncclCommInitAll + grouped ncclAllReduce(ncclInt64, ncclSum); librccl is resolved at run time
 *                (the copy the process already holds, else /opt/rocm's); distinct devices only.
 *
 * scv_comm_create with more than one rank -- or with SCV_COMM_RCCL and any rank count -- ends with a SELF-TEST (first contact with
 * the devices is loud, not a wrong accuracy later): two rounds of known int64 patterns -- (PEER) every rank reads every other rank's
 * buffer directly, ordered by the all-reduce's own cross-device events, once with nontemporal loads and once with ordinary loads,
 * then one whole scv_allreduce_counters of the production payload (8217 words) and one scv_allgather_i64, each verified on every
 * device.  The peer reads decide how this communicator reads its peers (nontemporal loads when they were right in both rounds,
 * else ordinary loads when those were; "peer_loads" says which); when neither kind was right, or a collective's result is wrong,
 * the create fails with SCV_ERR_ARG and a message ("... self-test ...") that names the device pair (peer read) or the device
 * (collective) -- the caller then creates the communicator with SCV_COMM_RCCL (the Python MultiDeviceEngine does that by
 * itself, with a warning).  The second round catches a reader that kept stale lines of a peer's first-round buffer.
 *
 * scv_allreduce_counters: buffers[r] is a DEVICE pointer on rank r's device (count int64 each, e.g. the packed counters
 * tie_class_hits | token_sum | truth_count_sum, optionally one more word from scv_export_error_word); in place, SUM, ordered
 * behind everything queued on the ranks' ctx streams, asynchronous.  Integer sums => the same bits at any device count.
 * The one-shot form stages through a buffer allocated at create (1 MiB per rank: no allocation on the launch path, legal
 * under hipGraph capture); a larger count grows it (synchronising), which is refused while a rank's stream is being captured.
 *
 * BASELINE config 5 (pass@k + bootstrap) on several GPUs without torch: the bootstrap resamples over ALL
 * problems, so after the vote every rank needs every rank's cells --
 * scv_allgather_cells: tables[r] is a DEVICE pointer on rank r's device to the WHOLE table scv_cell [sum(rows), B] in which rank r
 *   has written its own block (rows[r] problems, starting at row rows[0] + ... + rows[r - 1]: pass cells_out = that row to
 *   scv_aggregate_i32); afterwards every rank's table holds every block.  PEER: every rank pulls the other blocks over its own
 *   links (device-to-device copies ordered by the same events); RCCL: one grouped ncclBroadcast per block.
 * then every rank runs scv_bootstrap over its slice [r_begin, r_end) of the resamples, and
 * scv_allgather_i64: the same in-place all-gather for int64 blocks (counts[r] words per rank, rank r's block at word offset
 *   counts[0] + ... + counts[r - 1] of buffers[r]) -- the gather of the resample slices counts_out [r_end - r_begin, B, M].
 * Both are asynchronous and ordered behind everything queued on the ranks' ctx streams.
 * scv_comm_sync = scv_sync on every rank (first error wins).  scv_comm_get_stat: "selftest_words" (words verified per rank at
 * create; 0 with one SCV_COMM_PEER rank), "staging_bytes" (current size of the all-reduce's staging buffer), "peer_loads" (0: the
 * one-shot all-reduce reads its peers with nontemporal loads, 1: with ordinary loads, -1: no peer reads -- RCCL or one rank),
 * "selftest_nt_ok" / "selftest_plain_ok" (pairwise peer reads of the self-test with either load kind: 1 all right, 0 some wrong,
 * -1 not run).
 */
typedef struct scv_comm scv_comm;
#define SCV_COMM_PEER 0x0u
#define SCV_COMM_RCCL 0x1u
int scv_comm_create(scv_comm** out, const int* devices, int n, uint32_t ctx_flags, uint32_t comm_flags);
int scv_comm_destroy(scv_comm* comm);
int scv_comm_size(const scv_comm* comm);
scv_ctx* scv_comm_ctx(scv_comm* comm, int rank);       /* owned by the communicator: do not scv_destroy it */
int scv_allreduce_counters(scv_comm* comm, int64_t* const* buffers, int64_t count);
int scv_allgather_cells(scv_comm* comm, scv_cell* const* tables, const int64_t* rows, int32_t B);
int scv_allgather_i64(scv_comm* comm, int64_t* const* buffers, const int64_t* counts);
int scv_comm_sync(scv_comm* comm);
int scv_comm_get_stat(scv_comm* comm, const char* key, int64_t* out);

/* Duration of the most recent aggregation kernel launch on this ctx, from hipEvents recorded on
 * the launch stream (needs SCV_FLAG_TIMING).  Blocks until that launch has finished. */
int scv_last_kernel_ns(scv_ctx* ctx, uint64_t* ns_out);
/* Sum and count of all timed launches since the previous call (resets the accumulators). */
int scv_drain_kernel_ns(scv_ctx* ctx, uint64_t* total_ns_out, uint64_t* launches_out);

/*
 * Pinned (page-locked) host memory for HOST-mode inputs.  The reference keeps its samples in one JSON cache
 * (o1.py:50-68); an extractor that writes the answers / tokens tensors straight into a buffer from here lets
 * SCV_MEM_HOST calls DMA them in place (the staging pipeline otherwise copies pageable memory into pinned
 * bounce slots with worker threads first).  Any hipHostMalloc / hipHostRegister'ed buffer is recognised the
 * same way; these two entry points only spare a ctypes caller a HIP binding of its own.
 */
int scv_host_alloc(void** out, size_t bytes);
int scv_host_free(void* p);

/* How often this ctx took a form (monotonic counters, for tests and bench lines): "boot_fused" / "boot_separate"
 * (scv_aggregate_bootstrap_i32: one launch / two), "boot_cooperative" (one-launch forms started as cooperative launches),
 * "boot_recovered" (grid-barrier timeouts repaired by scv_sync with a separate bootstrap launch), "overwrite_fused" (counters
 * overwritten by the vote kernel's last workgroup), "lds_counters" (register-resident launches that produced their counters
 * themselves), "sort_cells" (sorted-cells launches), "few_votes" (launches of the kernels for cells of exactly 1, 2 or 4 votes), "one_vote" (those of them served by scv_one_vote / scv_two_votes: N = 1, 2), "prefix_cells" / "prefix_lane" / "prefix_pool" (prefix calls served by the cell kernels / by
 * the one-lane-per-problem kernel / by the one-pass-per-problem kernel), "prefix_sort" (launches of scv_sort_prefix: every power-of-two budget out of one
 * sort per problem -- queued, that is: a DEVICE-mode launch may find budgets it does not serve and leave them to the kernel behind it), "prefix_tokens" (launches of
 * scv_sort_prefix2<true>: the token sums of pools of 68 .. 128 votes out of token steps of the sort kernel's launch), "host_small_calls" / "host_pipelined_calls" (HOST-mode calls served by the one-block small path / by
 * the staging pipeline), "host_thread_start_failures" (worker threads of the staging pipeline the system refused to start: the
 * pipeline runs with the threads it has, the calling thread at least). */
int scv_get_stat(scv_ctx* ctx, const char* key, int64_t* out);

int scv_device_count(void);
/* Static properties of the ctx device: [0]=CU count, [1]=LDS bytes per workgroup max, [2]=clock kHz, [3]=HBM bytes. */
int scv_device_info(scv_ctx* ctx, int64_t info_out[4]);
const char* scv_last_error(void);   /* thread-local, never NULL */
const char* scv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SCVOTE_H */
