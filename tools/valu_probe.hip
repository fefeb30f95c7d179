// valu_probe.hip -- measurement utility (not product code): issue rate of the VALU instructions the one-lane-per-cell kernels are made of
// (scv_sort_cells: v_pk_min_u16 / v_pk_max_u16 / v_alignbit / v_bfi; its scan: v_pk_add_u16, v_pk_sub_u16 clamp, v_xor ...) on gfx950, so
// that "VALU busy" in the PMC summaries can be turned into a fraction of what the pipe can issue: cycles per wave-instruction on one
// SIMD with 1, 2 and 4 resident waves, for INDEPENDENT streams (16 accumulators) and for a DEPENDENT chain.
//
//   valu_probe.bin            table: instruction x {independent, dependent} x waves per SIMD -> cycles per instruction per SIMD
//
// Method: every wave runs REPS x 64 instructions between two s_memtime reads; the workgroup has 256 x W threads (W waves on each of the 4
// SIMDs of a CU), one workgroup per CU; cycles per instruction per SIMD = (t1 - t0) of a wave / (REPS x 64 x W) ... the median over waves.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kReps = 256;

// 16 independent accumulators, four instructions each per repetition
#define INDEP4(OP)                                                                                                                         \
    asm volatile(OP " %0, %0, %16\n" OP " %1, %1, %16\n" OP " %2, %2, %16\n" OP " %3, %3, %16\n" OP " %4, %4, %16\n" OP " %5, %5, %16\n"     \
                 OP " %6, %6, %16\n" OP " %7, %7, %16\n" OP " %8, %8, %16\n" OP " %9, %9, %16\n" OP " %10, %10, %16\n"                      \
                 OP " %11, %11, %16\n" OP " %12, %12, %16\n" OP " %13, %13, %16\n" OP " %14, %14, %16\n" OP " %15, %15, %16\n"              \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),  \
                   "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])                                            \
                 : "v"(k))
#define DEP16(OP)                                                                                                                          \
    asm volatile(OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n"           \
                 OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n"           \
                 OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n" OP " %0, %0, %1\n"                                               \
                 : "+v"(r[0]) : "v"(k))

// three-operand forms (v_alignbit_b32 d, a, b, 16; v_bfi_b32 d, mask, a, b; v_perm; v_or3; v_min3)
#define INDEP4_3(OP, C)                                                                                                                    \
    asm volatile(OP " %0, %0, %16, " C "\n" OP " %1, %1, %16, " C "\n" OP " %2, %2, %16, " C "\n" OP " %3, %3, %16, " C "\n"                 \
                 OP " %4, %4, %16, " C "\n" OP " %5, %5, %16, " C "\n" OP " %6, %6, %16, " C "\n" OP " %7, %7, %16, " C "\n"                 \
                 OP " %8, %8, %16, " C "\n" OP " %9, %9, %16, " C "\n" OP " %10, %10, %16, " C "\n" OP " %11, %11, %16, " C "\n"             \
                 OP " %12, %12, %16, " C "\n" OP " %13, %13, %16, " C "\n" OP " %14, %14, %16, " C "\n" OP " %15, %15, %16, " C "\n"         \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),  \
                   "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])                                            \
                 : "v"(k))
#define DEP16_3(OP, C)                                                                                                                     \
    asm volatile(OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n"                     \
                 OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n"                     \
                 OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n"                     \
                 OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n" OP " %0, %0, %1, " C "\n"                     \
                 : "+v"(r[0]) : "v"(k))

enum Op { PK_MIN_U16, PK_MAX_U16, PK_ADD_U16, PK_SUB_CLAMP, MIN_U32, XOR_B32, ADD_U32, ALIGNBIT, BFI, PERM, OR3, MIN3_U32, PAIR_MINMAX, CNDMASK,
          PK_MAD_U16, LSHL_OR, FMA_F32, CNDMASK_SGPR, CNDMASK_MIX, V_CMP, ADDC, CND_E32_SMOV, CND_E64_VCC, CND_E64_VCMP, kOps };
static const char* kNames[kOps] = {"v_pk_min_u16", "v_pk_max_u16", "v_pk_add_u16", "v_pk_sub_u16 clamp", "v_min_u32", "v_xor_b32", "v_add_u32",
                                   "v_alignbit_b32", "v_bfi_b32", "v_perm_b32", "v_or3_b32", "v_min3_u32", "pk_min+pk_max pair (an exchange)",
                                   "v_cndmask_b32 (vcc)", "v_pk_mad_u16", "v_lshl_or_b32", "v_fma_f32", "v_cndmask_b32 (SGPR pair, s_mov)",
                                   "v_cndmask_b32 + v_xor_b32 alternating", "v_cmp_gt_u32 (vcc)", "v_addc_co_u32 (vcc in / out)",
                                   "v_cndmask_b32_e32, vcc written by s_mov", "v_cndmask_b32_e64 ... vcc, vcc written by v_cmp", "v_cndmask_b32_e64 ... s[n:n+1] written by v_cmp"};

template <int OP, bool DEP>
__global__ void __launch_bounds__(1024) probe(long long* cycles, uint32_t* sink, uint32_t seed) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed * (threadIdx.x + 17u * i + 1u);
    const uint32_t k = seed ^ threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kReps; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == PK_MIN_U16) { if (DEP) DEP16("v_pk_min_u16"); else INDEP4("v_pk_min_u16"); }
            if (OP == PK_MAX_U16) { if (DEP) DEP16("v_pk_max_u16"); else INDEP4("v_pk_max_u16"); }
            if (OP == PK_ADD_U16) { if (DEP) DEP16("v_pk_add_u16"); else INDEP4("v_pk_add_u16"); }
            if (OP == PK_SUB_CLAMP) { if (DEP) DEP16_3("v_pk_sub_u16", "clamp"); else INDEP4_3("v_pk_sub_u16", "clamp"); }
            if (OP == MIN_U32) { if (DEP) DEP16("v_min_u32"); else INDEP4("v_min_u32"); }
            if (OP == XOR_B32) { if (DEP) DEP16("v_xor_b32"); else INDEP4("v_xor_b32"); }
            if (OP == ADD_U32) { if (DEP) DEP16("v_add_u32"); else INDEP4("v_add_u32"); }
            if (OP == ALIGNBIT) { if (DEP) DEP16_3("v_alignbit_b32", "16"); else INDEP4_3("v_alignbit_b32", "16"); }
            if (OP == BFI) { if (DEP) DEP16_3("v_bfi_b32", "%1"); else INDEP4_3("v_bfi_b32", "%16"); }
            if (OP == PERM) { if (DEP) DEP16_3("v_perm_b32", "%1"); else INDEP4_3("v_perm_b32", "%16"); }
            if (OP == OR3) { if (DEP) DEP16_3("v_or3_b32", "%1"); else INDEP4_3("v_or3_b32", "%16"); }
            if (OP == MIN3_U32) { if (DEP) DEP16_3("v_min3_u32", "%1"); else INDEP4_3("v_min3_u32", "%16"); }
            if (OP == PK_MAD_U16) { if (DEP) DEP16_3("v_pk_mad_u16", "%1"); else INDEP4_3("v_pk_mad_u16", "%16"); }
            if (OP == LSHL_OR) { if (DEP) DEP16_3("v_lshl_or_b32", "%1"); else INDEP4_3("v_lshl_or_b32", "%16"); }
            if (OP == FMA_F32) { if (DEP) DEP16_3("v_fma_f32", "%1"); else INDEP4_3("v_fma_f32", "%16"); }
            if (OP == CNDMASK) {
                // 16 selects on one vcc, in ONE statement (hipcc pads every asm statement that clobbers vcc with an s_nop)
                if (DEP) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\n"
                                      "v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\n"
                                      "v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\n"
                                      "v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[0]) : "v"(k) : "vcc");
                else asm volatile("v_cmp_gt_u32 vcc, %0, %16\n s_nop 1\n v_cndmask_b32 %0, %0, %16, vcc\nv_cndmask_b32 %1, %1, %16, vcc\nv_cndmask_b32 %2, %2, %16, vcc\nv_cndmask_b32 %3, %3, %16, vcc\n"
                                  "v_cndmask_b32 %4, %4, %16, vcc\nv_cndmask_b32 %5, %5, %16, vcc\nv_cndmask_b32 %6, %6, %16, vcc\nv_cndmask_b32 %7, %7, %16, vcc\n"
                                  "v_cndmask_b32 %8, %8, %16, vcc\nv_cndmask_b32 %9, %9, %16, vcc\nv_cndmask_b32 %10, %10, %16, vcc\nv_cndmask_b32 %11, %11, %16, vcc\n"
                                  "v_cndmask_b32 %12, %12, %16, vcc\nv_cndmask_b32 %13, %13, %16, vcc\nv_cndmask_b32 %14, %14, %16, vcc\nv_cndmask_b32 %15, %15, %16, vcc"
                                  : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                                    "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(k) : "vcc");
            }
            if (OP == CNDMASK_SGPR) {
                // the mask in an SGPR pair written by the scalar unit (no VALU -> SGPR hazard, no vcc)
                unsigned long long m;
                asm volatile("s_mov_b64 %0, 0x55555555" : "=s"(m));
                if (DEP) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(k), "s"(m));
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "s"(m));
                }
            }
            if (OP == CNDMASK_MIX) {
                unsigned long long m;
                asm volatile("s_mov_b64 %0, 0x55555555" : "=s"(m));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_cndmask_b32 %0, %0, %2, %3\n v_xor_b32 %1, %1, %2" : "+v"(r[DEP ? 0 : 2 * i]), "+v"(r[DEP ? 1 : 2 * i + 1]) : "v"(k), "s"(m));
            }
            if (OP == V_CMP) {
                asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %1\n v_cmp_gt_u32 vcc, %3, %1\n v_cmp_gt_u32 vcc, %4, %1\n"
                             "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %1\n v_cmp_gt_u32 vcc, %3, %1\n v_cmp_gt_u32 vcc, %4, %1\n"
                             "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %1\n v_cmp_gt_u32 vcc, %3, %1\n v_cmp_gt_u32 vcc, %4, %1\n"
                             "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %1\n v_cmp_gt_u32 vcc, %3, %1\n v_cmp_gt_u32 vcc, %4, %1"
                             : : "v"(r[0]), "v"(k), "v"(r[1]), "v"(r[2]), "v"(r[3]) : "vcc");
            }
            if (OP == ADDC) {
                if (DEP) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n"
                                      "v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n"
                                      "v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n"
                                      "v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc"
                                      : "+v"(r[0]) : "v"(k) : "vcc");
                else asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                                  "v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                                  "v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                                  "v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc"
                                  : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(k) : "vcc");
            }
            if (OP == CND_E32_SMOV) {
                asm volatile("s_mov_b64 vcc, 0x55555555" : : : "vcc");
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[DEP ? 0 : i]) : "v"(k) : "vcc");
            }
            if (OP == CND_E64_VCC) {
                asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_nop 3" : : "v"(r[0]), "v"(k) : "vcc");
#pragma unroll
                for (int i = DEP ? 0 : 1; i < 16; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r[DEP ? 0 : i]) : "v"(k) : "vcc");
            }
            if (OP == CND_E64_VCMP) {
                unsigned long long m;
                asm volatile("v_cmp_gt_u32 %0, %1, %2\n s_nop 3" : "=s"(m) : "v"(r[0]), "v"(k));
#pragma unroll
                for (int i = DEP ? 0 : 1; i < 16; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[DEP ? 0 : i]) : "v"(k), "s"(m));
            }
            if (OP == PAIR_MINMAX) {
                // one compare-exchange of the sorting network: (a, b) -> (min, max); 8 exchanges on 16 registers = 16 instructions
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int a = DEP ? 0 : 2 * i, b = DEP ? 1 : 2 * i + 1;
                    uint32_t t;
                    asm volatile("v_pk_min_u16 %0, %1, %2\n v_pk_max_u16 %2, %1, %2\n v_mov_b32 %1, %0" : "=&v"(t), "+v"(r[a]), "+v"(r[b]));
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= r[i];
    if (x == 0x1234567u) *sink = x;
    if ((threadIdx.x & 63) == 0) cycles[(long long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
static void run(long long* d_cycles, uint32_t* d_sink, int cus, double clock_ratio, double clock_khz) {
    for (int dep = 0; dep < 2; ++dep) {
        printf("%-34s %-11s", kNames[OP], dep ? "dependent" : "independent");
        for (int w : {1, 2, 4}) {
            const int threads = 256 * w, waves = cus * 4 * w;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (dep) hipLaunchKernelGGL((probe<OP, true>), dim3(cus), dim3(threads), 0, 0, d_cycles, d_sink, 12345u);
                else hipLaunchKernelGGL((probe<OP, false>), dim3(cus), dim3(threads), 0, 0, d_cycles, d_sink, 12345u);
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<long long> c(waves);
            CK(hipMemcpy(c.data(), d_cycles, sizeof(long long) * waves, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            // PAIR_MINMAX issues 24 instructions per 8 exchanges (min, max, mov); CNDMASK 16 selects + their compare
            const double per_rep = OP == PAIR_MINMAX ? 24.0 * 4 : (OP == CNDMASK ? 17.0 * 4 : 64.0);   // (CNDMASK_SGPR: + one s_mov_b64 per 16, not counted)
            const double cyc = (double)c[waves / 2] * clock_ratio / (kReps * per_rep * w);
            // cross-check against the wall clock of the launch (includes ~5 us of launch): cycles at the nominal shader clock
            const double ev = (double)ms * 1e-3 * clock_khz * 1e3 / (kReps * per_rep * w);
            printf("  W=%d: %6.2f (%5.2f by wall)", w, cyc, ev);
        }
        printf("   shader cycles per instruction per SIMD\n");
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    long long* d_cycles;
    uint32_t* d_sink;
    CK(hipMalloc(&d_cycles, sizeof(long long) * cus * 16));
    CK(hipMalloc(&d_sink, 4));
    // __builtin_readcyclecounter is s_memtime: a constant 100 MHz counter on gfx9 (not the shader clock); calibrate against wall time
    int wall_khz = 0, clk_khz = prop.clockRate;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    // s_memtime ticks at the shader clock on gfx950 (MI355X_MICROARCH.md: "tick = shader cycle"): ratio 1
    printf("device %s, %d CUs, clockRate %d kHz, wall clock %d kHz; cycles below are s_memtime ticks (= shader cycles on gfx950)\n", prop.name, cus,
           clk_khz, wall_khz);
    const double ratio = 1.0;
    run<PK_MIN_U16>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<PK_MAX_U16>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<PAIR_MINMAX>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<PK_ADD_U16>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<PK_SUB_CLAMP>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<PK_MAD_U16>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<MIN_U32>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<MIN3_U32>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<XOR_B32>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<ADD_U32>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<OR3>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<LSHL_OR>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<ALIGNBIT>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<BFI>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<PERM>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<CNDMASK>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<FMA_F32>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<CNDMASK_SGPR>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<CNDMASK_MIX>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<V_CMP>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<ADDC>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<CND_E32_SMOV>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<CND_E64_VCC>(d_cycles, d_sink, cus, ratio, clk_khz);
    run<CND_E64_VCMP>(d_cycles, d_sink, cus, ratio, clk_khz);
    return 0;
}
