/* tools/ubench/valu_issue.hip -- what the gfx950 SIMDs sustain in wave64 VALU instructions, opcode by opcode and for
 * the instruction mix of k_trace8's node step, at 1 / 2 / 5 / 8 waves per SIMD on every CU.
 *
 * The question: bench.py priced k_trace8 against one wave64 VALU instruction per SIMD per
 * FOUR clocks and the kernel's own SQ_INSTS_VALU came out at 100-104 % of that; MI355X_MICROARCH.md gives TWO
 * clocks on the SIMD-32 (four only for a wave alone on its SIMD).  This program measures it per opcode, because
 * the answer turned out to depend on the opcode.
 *
 * Geometry: 256-thread workgroups (one wave per SIMD each), k workgroups per CU forced by dynamic LDS
 * (120 / k LDS granules of 1280 bytes each), grid = CUs x k, so every SIMD holds exactly k waves.  Every wave executes kIter x 64
 * instructions on 8 rotating destination registers (no dependent-issue stall), written in inline asm so that
 * nothing is folded away; a launch lasts milliseconds, so ramp and tail do not matter.  Reported per class and k:
 *   g_wave_instructions_per_s   chip-wide, from the host's HIP-event time (what a roofline needs)
 *   clock_ghz                   in-kernel: delta(s_memtime) / delta(s_memrealtime) x 100 MHz, median over waves
 *   cycles_per_instruction_simd kernel time x clock / (k x instructions per wave)
 *
 * build:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -o valu_issue valu_issue.hip
 * run:    ./valu_issue > valu_issue.jsonl        (one JSON object per line)
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

/* dynamic LDS per workgroup such that exactly k workgroups fit a CU: LDS is handed out in 1280-byte granules (128 per CU);
 * k x floor(120 / k) granules fit with room to spare, k + 1 such workgroups do not (a first version asked for 160 KiB / k - 64
 * bytes, which rounds UP to a granule: only 4 of "5" workgroups were resident and the fifth ran alone afterwards) */
static size_t lds_bytes_for(int k) { return (size_t)(120 / k) * 1280; }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIter = 8192;
typedef void (*kern_t)(float*, unsigned long long*, float);

#define X8(S) S S S S S S S S

#define PROLOGUE                                                                                                       \
    extern __shared__ char lds_pad[];                                                                                  \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    float x = seed * 0.5f + 0.25f, y = 1.0f - seed * 1e-6f;                                                            \
    uint32_t u0 = __float_as_uint(seed) | 1u, u1 = u0 * 3u, u2 = u0 * 5u, u3 = u0 * 7u, u4 = u0 * 11u, u5 = u0 * 13u, u6 = u0 * 17u, u7 = u0 * 19u; \
    uint32_t sel = 0x04010400u, w = 0x64646464u;                                                                       \
    (void)x; (void)y; (void)sel; (void)w;                                                                              \
    if (threadIdx.x == 1023) lds_pad[0] = 1;     /* keeps the dynamic allocation alive; never true */                  \
    __syncthreads();                                                                                                   \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                 \
    for (int it = 0; it < kIter; it++) {
#define EPILOGUE                                                                                                       \
    }                                                                                                                  \
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();               \
    if ((threadIdx.x & 63) == 0) {                                                                                     \
        const size_t wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                                \
        cycles[2 * wv] = t1_ - t0; cycles[2 * wv + 1] = r1_ - r0;                                                      \
    }                                                                                                                  \
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + __uint_as_float((u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) & 0x3fffffffu); \
    if (s == 123.456f) sink[0] = s;        /* never true in practice; keeps every chain live */

/* dst = OP(x, y, dst) on float registers */
#define DEF_F3(NAME, OP, SUF)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* sink, unsigned long long* cycles, float seed) {                 \
        PROLOGUE                                                                                                       \
        X8(asm volatile(OP " %0, %8, %9, %0 " SUF "\n" OP " %1, %8, %9, %1 " SUF "\n" OP " %2, %8, %9, %2 " SUF "\n" OP " %3, %8, %9, %3 " SUF "\n" \
                        OP " %4, %8, %9, %4 " SUF "\n" OP " %5, %8, %9, %5 " SUF "\n" OP " %6, %8, %9, %6 " SUF "\n" OP " %7, %8, %9, %7 " SUF      \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)                    \
        EPILOGUE                                                                                                       \
    }
/* dst = OP(w, sel, dst) on integer registers */
#define DEF_U3(NAME, OP)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(float* sink, unsigned long long* cycles, float seed) {                 \
        PROLOGUE                                                                                                       \
        X8(asm volatile(OP " %0, %8, %9, %0\n" OP " %1, %8, %9, %1\n" OP " %2, %8, %9, %2\n" OP " %3, %8, %9, %3\n"     \
                        OP " %4, %8, %9, %4\n" OP " %5, %8, %9, %5\n" OP " %6, %8, %9, %6\n" OP " %7, %8, %9, %7"       \
                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w), "v"(sel));)                  \
        EPILOGUE                                                                                                       \
    }
/* dst = OP(x, dst) (VOP2) on float registers */
#define DEF_F2(NAME, OP)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(float* sink, unsigned long long* cycles, float seed) {                 \
        PROLOGUE                                                                                                       \
        X8(asm volatile(OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n"                     \
                        OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7"                       \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)   \
        EPILOGUE                                                                                                       \
    }
#define DEF_U2(NAME, OP)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(float* sink, unsigned long long* cycles, float seed) {                 \
        PROLOGUE                                                                                                       \
        X8(asm volatile(OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n"                     \
                        OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7"                       \
                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(sel));) \
        EPILOGUE                                                                                                       \
    }
/* dst = OP(dst) (VOP1), with an optional suffix (DPP / SDWA controls) */
#define DEF_1(NAME, OP, SUF, R0, R1, R2, R3, R4, R5, R6, R7)                                                          \
    __global__ __launch_bounds__(256) void NAME(float* sink, unsigned long long* cycles, float seed) {                 \
        PROLOGUE                                                                                                       \
        X8(asm volatile(OP " %0, %0 " SUF "\n" OP " %1, %1 " SUF "\n" OP " %2, %2 " SUF "\n" OP " %3, %3 " SUF "\n"      \
                        OP " %4, %4 " SUF "\n" OP " %5, %5 " SUF "\n" OP " %6, %6 " SUF "\n" OP " %7, %7 " SUF          \
                        : "+v"(R0), "+v"(R1), "+v"(R2), "+v"(R3), "+v"(R4), "+v"(R5), "+v"(R6), "+v"(R7));)            \
        EPILOGUE                                                                                                       \
    }
#define DEF_1F(NAME, OP, SUF) DEF_1(NAME, OP, SUF, a0, a1, a2, a3, a4, a5, a6, a7)
#define DEF_1U(NAME, OP, SUF) DEF_1(NAME, OP, SUF, u0, u1, u2, u3, u4, u5, u6, u7)

DEF_F3(k_fma_f32, "v_fma_f32", "")
DEF_F3(k_fma_mix_lo, "v_fma_mix_f32", "op_sel_hi:[1,0,0]")
DEF_F3(k_fma_mix_hi, "v_fma_mix_f32", "op_sel:[1,0,0] op_sel_hi:[1,0,0]")
DEF_F3(k_max3_f32, "v_max3_f32", "")
DEF_F3(k_min3_f32, "v_min3_f32", "")
DEF_F3(k_med3_f32, "v_med3_f32", "")
DEF_F3(k_fma_f16, "v_fma_f16", "")
DEF_F3(k_pk_fma_f16, "v_pk_fma_f16", "")
DEF_U3(k_perm_b32, "v_perm_b32")
DEF_U3(k_and_or_b32, "v_and_or_b32")
DEF_U3(k_lshl_add_u32, "v_lshl_add_u32")
DEF_U3(k_lshl_or_b32, "v_lshl_or_b32")
DEF_U3(k_add3_u32, "v_add3_u32")
DEF_U3(k_bfe_u32, "v_bfe_u32")
DEF_U3(k_bfi_b32, "v_bfi_b32")
DEF_U3(k_alignbit_b32, "v_alignbit_b32")
DEF_U3(k_mad_u32_u24, "v_mad_u32_u24")
DEF_U3(k_xad_u32, "v_xad_u32")
DEF_U3(k_or3_b32, "v_or3_b32")
DEF_F2(k_add_f32, "v_add_f32")
DEF_F2(k_mul_f32, "v_mul_f32")
DEF_F2(k_sub_f32, "v_sub_f32")
DEF_F2(k_max_f32, "v_max_f32")
DEF_F2(k_min_f32, "v_min_f32")
DEF_F2(k_mul_legacy, "v_mul_legacy_f32")
DEF_U2(k_and_b32, "v_and_b32")
DEF_U2(k_or_b32, "v_or_b32")
DEF_U2(k_xor_b32, "v_xor_b32")
DEF_U2(k_add_u32, "v_add_u32")
DEF_U2(k_sub_u32, "v_sub_u32")
DEF_U2(k_lshlrev_b32, "v_lshlrev_b32")
DEF_U2(k_lshrrev_b32, "v_lshrrev_b32")
DEF_U2(k_mul_u32_u24, "v_mul_u32_u24")
DEF_U2(k_mul_lo_u32, "v_mul_lo_u32")
DEF_U2(k_max_u32, "v_max_u32")
DEF_1U(k_mov_b32, "v_mov_b32", "")
DEF_1U(k_not_b32, "v_not_b32", "")
DEF_1U(k_bfrev_b32, "v_bfrev_b32", "")
DEF_1U(k_ffbh_u32, "v_ffbh_u32", "")
DEF_1U(k_mov_dpp_quad, "v_mov_b32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_1U(k_mov_dpp_row_shr, "v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_1F(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0", "")
DEF_1F(k_cvt_f32_ubyte3, "v_cvt_f32_ubyte3", "")
DEF_1F(k_cvt_f32_u32, "v_cvt_f32_u32", "")
DEF_1F(k_cvt_f32_f16, "v_cvt_f32_f16", "")
DEF_1F(k_cvt_f32_u32_sdwa, "v_cvt_f32_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")
DEF_1F(k_rcp_f32, "v_rcp_f32", "")
DEF_1F(k_sqrt_f32, "v_sqrt_f32", "")

__global__ __launch_bounds__(256) void k_fmac_f32(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                    "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_max_lit(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_max_f32 %0, 0xb8d1bc76, %0\n v_max_f32 %1, 0xb8d1bc76, %1\n v_max_f32 %2, 0xb8d1bc76, %2\n v_max_f32 %3, 0xb8d1bc76, %3\n"
                    "v_max_f32 %4, 0xb8d1bc76, %4\n v_max_f32 %5, 0xb8d1bc76, %5\n v_max_f32 %6, 0xb8d1bc76, %6\n v_max_f32 %7, 0xb8d1bc76, %7"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cmp_addc(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_cmp_le_f32 vcc, %4, %5\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %5, %4\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n"
                    "v_cmp_le_f32 vcc, %4, %5\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_le_f32 vcc, %5, %4\n v_addc_co_u32 %3, vcc, %3, %3, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(x), "v"(y) : "vcc");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cmp_vcc(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n"
                    "v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0"
                    : : "v"(x), "v"(y) : "vcc");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cmp_sgpr(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n v_cmp_le_f32 s[22:23], %1, %0\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %1, %0\n"
                    "v_cmp_le_f32 s[20:21], %0, %1\n v_cmp_le_f32 s[22:23], %1, %0\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %1, %0"
                    : : "v"(x), "v"(y) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    EPILOGUE
}
/* v_cndmask_b32 in its forms: VOP2 with a VCC that nothing rewrites, VOP3 with an SGPR-pair mask, VOP2 behind a v_cmp */
__global__ __launch_bounds__(256) void k_cndmask_vcc_stale(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                    "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w) : "vcc");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cndmask_vcc_smov(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
    X8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                    "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w) : "vcc");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cndmask_sgpr(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555" ::: "s20", "s21");
    X8(asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                    "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w) : "s20", "s21");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cndmask_distinct(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
    X8(asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                    "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
                    : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w), "v"(sel) : "vcc");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cmp_cndmask(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %0, %0, %6, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %1, %1, %6, vcc\n"
                    "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %2, %2, %6, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %3, %3, %6, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(x), "v"(y), "v"(w) : "vcc");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_pk_fma_f32(float* sink, unsigned long long* cycles, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7};
    PROLOGUE
    f2 px = {x, x}, py = {y, y};
    X8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                    "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px), "v"(py));)
    a0 += p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_pk_mul_f32(float* sink, unsigned long long* cycles, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7};
    PROLOGUE
    f2 py = {y, y};
    X8(asm volatile("v_pk_mul_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n"
                    "v_pk_mul_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3"
                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(py));)
    a0 += p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    EPILOGUE
}

/* how far from the v_cmp that wrote VCC does a VOP2 v_cndmask stay fast? */
__global__ __launch_bounds__(256) void k_cmp_cndmask7(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n"
                    "v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6) : "v"(x), "v"(y), "v"(w) : "vcc");)
    EPILOGUE
}
/* the shape of k_trace8's node step: one v_cmp -> vcc, two v_cmp -> SGPR pairs, then VOP2 selects on vcc with other work between */
__global__ __launch_bounds__(256) void k_cmp_gap_cndmask(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_cmp_gt_f32 vcc, 0, %4\n v_cmp_gt_f32 s[20:21], 0, %5\n v_cmp_gt_f32 s[22:23], 0, %4\n v_cndmask_b32 %0, %0, %6, vcc\n"
                    "v_and_b32 %2, %6, %2\n v_and_b32 %3, %6, %3\n v_cndmask_b32 %1, %1, %6, vcc\n v_and_b32 %2, %6, %2"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(x), "v"(y), "v"(w) : "vcc", "s20", "s21", "s22", "s23");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_cndmask_e64_vcc(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
    X8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
                    "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w) : "vcc");)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_bfi_select(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_bfi_b32 %0, %9, %8, %0\n v_bfi_b32 %1, %9, %8, %1\n v_bfi_b32 %2, %9, %8, %2\n v_bfi_b32 %3, %9, %8, %3\n"
                    "v_bfi_b32 %4, %9, %8, %4\n v_bfi_b32 %5, %9, %8, %5\n v_bfi_b32 %6, %9, %8, %6\n v_bfi_b32 %7, %9, %8, %7"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w), "v"(sel));)
    EPILOGUE
}
__global__ __launch_bounds__(256) void k_ashr_pk(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    X8(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_pk_max_f16 %1, %8, %1\n v_pk_min_f16 %2, %8, %2\n v_pk_add_f16 %3, %8, %3\n"
                    "v_ashrrev_i32 %4, 31, %4\n v_pk_max_f16 %5, %8, %5\n v_pk_min_f16 %6, %8, %6\n v_pk_mul_f16 %7, %8, %7"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w));)
    EPILOGUE
}
DEF_U2(k_ashrrev_i32, "v_ashrrev_i32")
DEF_U2(k_pk_max_f16, "v_pk_max_f16")
DEF_U2(k_pk_add_f16, "v_pk_add_f16")
DEF_U2(k_bcnt, "v_bcnt_u32_b32")

/* the instruction stream of SLAB8x4 in kernels.hip: 6 v_perm + slab8<HI> + slab8<LO>, 6 v_perm + 2 slab8 = 60 instructions */
__global__ __launch_bounds__(256) void k_slab8x4(float* sink, unsigned long long* cycles, float seed) {
    PROLOGUE
    uint32_t raw = u0;
    float tn, tf, t1, t2;
    uint32_t c0, c1, c2, c3, c4, c5;
#define UB_SLAB(SEL)                                                                                               \
    asm volatile("v_fma_mix_f32 %[t1], %[c0], %[x], %[a0] " SEL "\n v_fma_mix_f32 %[t2], %[c2], %[y], %[a2] " SEL "\n"  \
                 "v_fma_mix_f32 %[tn], %[c4], %[x], %[a4] " SEL "\n v_max_f32 %[tn], 0xb8d1bc76, %[tn]\n v_max3_f32 %[tn], %[t1], %[t2], %[tn]\n" \
                 "v_fma_mix_f32 %[t1], %[c1], %[x], %[a1] " SEL "\n v_fma_mix_f32 %[t2], %[c3], %[y], %[a3] " SEL "\n"  \
                 "v_fma_mix_f32 %[tf], %[c5], %[x], %[a5] " SEL "\n v_min_f32 %[tf], %[tf], %[a6]\n v_min3_f32 %[tf], %[t1], %[t2], %[tf]\n" \
                 "v_cmp_le_f32 vcc, %[tn], %[tf]\n v_addc_co_u32 %[raw], vcc, %[raw], %[raw], vcc"                    \
                 : [tn] "=&v"(tn), [tf] "=&v"(tf), [t1] "=&v"(t1), [t2] "=&v"(t2), [raw] "+v"(raw)                     \
                 : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4), [c5] "v"(c5), [x] "v"(x), [y] "v"(y), \
                   [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5), [a6] "v"(a6)    \
                 : "vcc");
#define UB_PERM6(S)                                                                                                \
    asm volatile("v_perm_b32 %0, %6, %7, %13\n v_perm_b32 %1, %6, %8, %13\n v_perm_b32 %2, %6, %9, %13\n"           \
                 "v_perm_b32 %3, %6, %10, %13\n v_perm_b32 %4, %6, %11, %13\n v_perm_b32 %5, %6, %12, %13"          \
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4), "=&v"(c5)                                  \
                 : "v"(w), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(S));
    UB_PERM6(sel)
    UB_SLAB("op_sel:[1,0,0] op_sel_hi:[1,0,0]")
    UB_SLAB("op_sel_hi:[1,0,0]")
    UB_PERM6(u7)
    UB_SLAB("op_sel:[1,0,0] op_sel_hi:[1,0,0]")
    UB_SLAB("op_sel_hi:[1,0,0]")
    u0 = raw;
    EPILOGUE
}

struct Cls { const char* name; kern_t fn; int body; };

static void run_class(const Cls& c, int cus, float* sink, unsigned long long* d_cycles, std::vector<unsigned long long>& h) {
    const int ks[] = {1, 2, 5, 8};
    for (int k : ks) {
        const size_t lds = lds_bytes_for(k);
        CHECK(hipFuncSetAttribute((const void*)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int blocks = cus * k;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float best_ms = 1e30f;
        for (int rep = 0; rep < 3; rep++) {      /* the first repetition warms the clock; keep the best of the rest */
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), lds, 0, sink, d_cycles, 1.0f + rep);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best_ms) best_ms = ms;
        }
        const int nw = blocks * 4;
        CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost));
        std::vector<double> clk(nw), cyc(nw);
        for (int i = 0; i < nw; i++) { cyc[i] = (double)h[2 * i]; clk[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.; }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        const double clock_ghz = clk[nw / 2];
        const double instr = (double)kIter * c.body;
        const double cpi = best_ms * 1e-3 * clock_ghz * 1e9 / (k * instr);
        printf("{\"bench\": \"valu_issue\", \"class\": \"%s\", \"waves_per_simd\": %d, \"wave_instructions_per_wave\": %.0f, "
               "\"kernel_ms\": %.4f, \"g_wave_instructions_per_s\": %.1f, \"clock_ghz\": %.3f, \"cycles_per_instruction_simd\": %.3f, "
               "\"median_cycles_per_wave\": %.0f}\n",
               c.name, k, instr, best_ms, instr * nw / (best_ms * 1e-3) / 1e9, clock_ghz, cpi, cyc[nw / 2]);
        fflush(stdout);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    fprintf(stderr, "%s: %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    float* sink; unsigned long long* d_cycles;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * 2 * cus * 8 * 4));
    std::vector<unsigned long long> h((size_t)2 * cus * 8 * 4);
    const Cls classes[] = {
        {"k_trace8 node step (12 v_perm + 4 x 12-instruction slab test)", k_slab8x4, 60},
        {"v_fma_f32", k_fma_f32, 64}, {"v_fmac_f32", k_fmac_f32, 64}, {"v_add_f32", k_add_f32, 64}, {"v_mul_f32", k_mul_f32, 64},
        {"v_sub_f32", k_sub_f32, 64}, {"v_max_f32", k_max_f32, 64}, {"v_min_f32", k_min_f32, 64}, {"v_mul_legacy_f32", k_mul_legacy, 64},
        {"v_max_f32 with a 32-bit literal", k_max_lit, 64},
        {"v_fma_mix_f32 (lo half)", k_fma_mix_lo, 64}, {"v_fma_mix_f32 (hi half)", k_fma_mix_hi, 64},
        {"v_max3_f32", k_max3_f32, 64}, {"v_min3_f32", k_min3_f32, 64}, {"v_med3_f32", k_med3_f32, 64},
        {"v_fma_f16", k_fma_f16, 64}, {"v_pk_fma_f16", k_pk_fma_f16, 64}, {"v_pk_fma_f32", k_pk_fma_f32, 64}, {"v_pk_mul_f32 / v_pk_add_f32", k_pk_mul_f32, 64},
        {"v_perm_b32", k_perm_b32, 64}, {"v_and_or_b32", k_and_or_b32, 64}, {"v_lshl_add_u32", k_lshl_add_u32, 64}, {"v_lshl_or_b32", k_lshl_or_b32, 64},
        {"v_add3_u32", k_add3_u32, 64}, {"v_bfe_u32", k_bfe_u32, 64}, {"v_bfi_b32", k_bfi_b32, 64}, {"v_alignbit_b32", k_alignbit_b32, 64},
        {"v_mad_u32_u24", k_mad_u32_u24, 64}, {"v_xad_u32", k_xad_u32, 64}, {"v_or3_b32", k_or3_b32, 64},
        {"v_and_b32", k_and_b32, 64}, {"v_or_b32", k_or_b32, 64}, {"v_xor_b32", k_xor_b32, 64}, {"v_add_u32", k_add_u32, 64}, {"v_sub_u32", k_sub_u32, 64},
        {"v_lshlrev_b32", k_lshlrev_b32, 64}, {"v_lshrrev_b32", k_lshrrev_b32, 64}, {"v_mul_u32_u24", k_mul_u32_u24, 64}, {"v_mul_lo_u32", k_mul_lo_u32, 64},
        {"v_max_u32", k_max_u32, 64},
        {"v_mov_b32", k_mov_b32, 64}, {"v_not_b32", k_not_b32, 64}, {"v_bfrev_b32", k_bfrev_b32, 64}, {"v_ffbh_u32", k_ffbh_u32, 64},
        {"v_mov_b32_dpp quad_perm", k_mov_dpp_quad, 64}, {"v_mov_b32_dpp row_shr:1", k_mov_dpp_row_shr, 64},
        {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte0, 64}, {"v_cvt_f32_ubyte3", k_cvt_f32_ubyte3, 64}, {"v_cvt_f32_u32", k_cvt_f32_u32, 64},
        {"v_cvt_f32_f16", k_cvt_f32_f16, 64}, {"v_cvt_f32_u32_sdwa src0_sel:BYTE_1", k_cvt_f32_u32_sdwa, 64},
        {"v_rcp_f32", k_rcp_f32, 64}, {"v_sqrt_f32", k_sqrt_f32, 64},
        {"v_cmp_le_f32 vcc + v_addc_co_u32", k_cmp_addc, 64}, {"v_cmp_l[et]_f32 -> vcc", k_cmp_vcc, 64}, {"v_cmp_l[et]_f32 -> SGPR pair (VOP3)", k_cmp_sgpr, 64},
        {"v_cndmask_b32 vcc (VOP2), vcc last written by a v_cmp long ago", k_cndmask_vcc_stale, 64},
        {"v_cndmask_b32 vcc (VOP2), vcc written by s_mov_b64", k_cndmask_vcc_smov, 64},
        {"v_cndmask_b32 SGPR-pair mask (VOP3)", k_cndmask_sgpr, 64},
        {"v_cndmask_b32 vcc (VOP2), destination distinct from sources", k_cndmask_distinct, 64},
        {"v_cmp_lt_f32 vcc + v_cndmask_b32 vcc", k_cmp_cndmask, 64},
        {"v_cmp_lt_f32 vcc + 7 x v_cndmask_b32 vcc", k_cmp_cndmask7, 64},
        {"k_trace8 select shape: v_cmp vcc, 2 v_cmp SGPR, v_cndmask vcc, 2 v_and, v_cndmask vcc, v_and", k_cmp_gap_cndmask, 64},
        {"v_cndmask_b32_e64 with vcc as its mask operand (VOP3), vcc written by s_mov", k_cndmask_e64_vcc, 64},
        {"v_bfi_b32 as a select", k_bfi_select, 64},
        {"v_ashrrev_i32", k_ashrrev_i32, 64}, {"v_pk_max_f16", k_pk_max_f16, 64}, {"v_pk_add_f16", k_pk_add_f16, 64}, {"v_bcnt_u32_b32", k_bcnt, 64},
    };
    const char* only = argc > 1 ? argv[1] : nullptr;
    for (const Cls& c : classes) {
        if (only && !strstr(c.name, only)) continue;
        run_class(c, cus, sink, d_cycles, h);
    }
    return 0;
}
