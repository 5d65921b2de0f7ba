// valu_patterns.hip -- issue cost of specific operand/dependency shapes on gfx950, written with
// explicit physical registers (the compiler's allocation is out of the picture).  Each pattern is an
// 8-instruction group repeated 16x per loop iteration; 8 waves/SIMD on every CU.
// Output: cycles per wave-instruction per SIMD at the nominal 2.4 GHz clock.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CLOBBERS "scc","s30","s31","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","vcc","s20","s21","s22","s23","s24","s25","s26","s27"
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R16(x) R4(x) R4(x) R4(x) R4(x)

#define PATTERN_KERNEL(NAME, GROUP)                                                        \
    __global__ void NAME(unsigned* out, int iters, unsigned seed)                         \
    {                                                                                      \
        asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %0\n"   \
                     "v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n"   \
                     "v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n" \
                     "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n" \
                     "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n" \
                     "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n" \
                     "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n" \
                     "s_mov_b32 s20, %1\n s_mov_b32 s21, %1\n s_mov_b32 s22, 0xD2511F53\n s_mov_b32 s23, 0xCD9E8D57\n" \
                     :: "v"(threadIdx.x * 2654435761u + seed), "s"(seed * 77u + 12345u) : CLOBBERS);  \
        for (int it = 0; it < iters; it++) { asm volatile(R16(GROUP) ::: CLOBBERS); }     \
        unsigned r;                                                                        \
        asm volatile("v_xor_b32 %0, v0, v1\n v_xor_b32 %0, %0, v2\n v_xor_b32 %0, %0, v3\n v_xor_b32 %0, %0, v16\n v_xor_b32 %0, %0, v18" : "=v"(r) :: CLOBBERS); \
        if (r == 0x12345679u) out[0] = r;                                                  \
    }

// --- xor: independent instructions, source operands in the same VGPR bank (index mod 4) or not
PATTERN_KERNEL(k_xor_samebank, "v_xor_b32 v0, v4, v8\n v_xor_b32 v1, v5, v9\n v_xor_b32 v2, v6, v10\n v_xor_b32 v3, v7, v11\n"
                               "v_xor_b32 v0, v8, v12\n v_xor_b32 v1, v9, v13\n v_xor_b32 v2, v10, v14\n v_xor_b32 v3, v11, v15\n")
PATTERN_KERNEL(k_xor_diffbank, "v_xor_b32 v0, v4, v9\n v_xor_b32 v1, v5, v10\n v_xor_b32 v2, v6, v11\n v_xor_b32 v3, v7, v8\n"
                               "v_xor_b32 v0, v8, v13\n v_xor_b32 v1, v9, v14\n v_xor_b32 v2, v10, v15\n v_xor_b32 v3, v11, v12\n")
PATTERN_KERNEL(k_xor_sgpr,     "v_xor_b32 v0, s20, v8\n v_xor_b32 v1, s21, v9\n v_xor_b32 v2, s20, v10\n v_xor_b32 v3, s21, v11\n"
                               "v_xor_b32 v0, s20, v12\n v_xor_b32 v1, s21, v13\n v_xor_b32 v2, s20, v14\n v_xor_b32 v3, s21, v15\n")
PATTERN_KERNEL(k_xor_literal,  "v_xor_b32 v0, 0x9E3779B9, v8\n v_xor_b32 v1, 0x9E3779B9, v9\n v_xor_b32 v2, 0x9E3779B9, v10\n v_xor_b32 v3, 0x9E3779B9, v11\n"
                               "v_xor_b32 v0, 0x9E3779B9, v12\n v_xor_b32 v1, 0x9E3779B9, v13\n v_xor_b32 v2, 0x9E3779B9, v14\n v_xor_b32 v3, 0x9E3779B9, v15\n")
// dependent chains: every instruction reads the previous result (distance 1) / 2 chains / 4 chains
PATTERN_KERNEL(k_xor_dep1,     "v_xor_b32 v0, v0, v8\n v_xor_b32 v0, v0, v9\n v_xor_b32 v0, v0, v10\n v_xor_b32 v0, v0, v11\n"
                               "v_xor_b32 v0, v0, v12\n v_xor_b32 v0, v0, v13\n v_xor_b32 v0, v0, v14\n v_xor_b32 v0, v0, v15\n")
PATTERN_KERNEL(k_xor_dep2,     "v_xor_b32 v0, v0, v8\n v_xor_b32 v1, v1, v9\n v_xor_b32 v0, v0, v10\n v_xor_b32 v1, v1, v11\n"
                               "v_xor_b32 v0, v0, v12\n v_xor_b32 v1, v1, v13\n v_xor_b32 v0, v0, v14\n v_xor_b32 v1, v1, v15\n")
// --- mad_u64: independent, SGPR multiplier, as the Philox loop uses it
PATTERN_KERNEL(k_mad_indep,    "v_mad_u64_u32 v[0:1], s[24:25], v8, s22, 0\n v_mad_u64_u32 v[2:3], s[24:25], v9, s23, 0\n v_mad_u64_u32 v[4:5], s[24:25], v10, s22, 0\n v_mad_u64_u32 v[6:7], s[24:25], v11, s23, 0\n"
                               "v_mad_u64_u32 v[0:1], s[24:25], v12, s22, 0\n v_mad_u64_u32 v[2:3], s[24:25], v13, s23, 0\n v_mad_u64_u32 v[4:5], s[24:25], v14, s22, 0\n v_mad_u64_u32 v[6:7], s[24:25], v15, s23, 0\n")
PATTERN_KERNEL(k_mad_dep2,     "v_mad_u64_u32 v[0:1], s[24:25], v1, s22, 0\n v_mad_u64_u32 v[2:3], s[24:25], v3, s23, 0\n v_mad_u64_u32 v[0:1], s[24:25], v1, s22, 0\n v_mad_u64_u32 v[2:3], s[24:25], v3, s23, 0\n"
                               "v_mad_u64_u32 v[0:1], s[24:25], v1, s22, 0\n v_mad_u64_u32 v[2:3], s[24:25], v3, s23, 0\n v_mad_u64_u32 v[0:1], s[24:25], v1, s22, 0\n v_mad_u64_u32 v[2:3], s[24:25], v3, s23, 0\n")
PATTERN_KERNEL(k_mulhi_mullo,  "v_mul_hi_u32 v0, v8, s22\n v_mul_lo_u32 v1, v8, s22\n v_mul_hi_u32 v2, v9, s23\n v_mul_lo_u32 v3, v9, s23\n"
                               "v_mul_hi_u32 v4, v10, s22\n v_mul_lo_u32 v5, v10, s22\n v_mul_hi_u32 v6, v11, s23\n v_mul_lo_u32 v7, v11, s23\n")
// --- one exact Philox round (2 mads + 4 xors, 2 keyed by SGPR), dependent round to round, as 2 rounds per group
#define PHILOX_ROUND_S "v_mad_u64_u32 v[4:5], s[24:25], v0, s22, 0\n v_mad_u64_u32 v[6:7], s[24:25], v2, s23, 0\n" \
                       "v_xor_b32 v0, v7, v1\n v_xor_b32 v2, v5, v3\n v_xor_b32 v0, s20, v0\n v_xor_b32 v2, s21, v2\n v_mov_b32 v1, v6\n v_mov_b32 v3, v4\n"
PATTERN_KERNEL(k_philox_round_sgprkey, PHILOX_ROUND_S)
#define PHILOX_ROUND_V "v_mad_u64_u32 v[4:5], s[24:25], v0, s22, 0\n v_mad_u64_u32 v[6:7], s[24:25], v2, s23, 0\n" \
                       "v_xor_b32 v0, v7, v1\n v_xor_b32 v2, v5, v3\n v_xor_b32 v0, v16, v0\n v_xor_b32 v2, v18, v2\n v_mov_b32 v1, v6\n v_mov_b32 v3, v4\n"
PATTERN_KERNEL(k_philox_round_vgprkey, PHILOX_ROUND_V)
#define PHILOX_ROUND_L "v_mad_u64_u32 v[4:5], s[24:25], v0, s22, 0\n v_mad_u64_u32 v[6:7], s[24:25], v2, s23, 0\n" \
                       "v_xor_b32 v0, v7, v1\n v_xor_b32 v2, v5, v3\n v_xor_b32 v0, 0x9E3779B9, v0\n v_xor_b32 v2, 0xBB67AE85, v2\n v_mov_b32 v1, v6\n v_mov_b32 v3, v4\n"
PATTERN_KERNEL(k_philox_round_literalkey, PHILOX_ROUND_L)
// two interleaved independent Philox streams (registers 0-7 and 8-15): same instruction count per stream
#define PHILOX_ROUND_2X "v_mad_u64_u32 v[4:5], s[24:25], v0, s22, 0\n v_mad_u64_u32 v[12:13], s[24:25], v8, s22, 0\n v_mad_u64_u32 v[6:7], s[24:25], v2, s23, 0\n v_mad_u64_u32 v[14:15], s[24:25], v10, s23, 0\n" \
                        "v_xor_b32 v0, v7, v1\n v_xor_b32 v8, v15, v9\n v_xor_b32 v2, v5, v3\n v_xor_b32 v10, v13, v11\n v_xor_b32 v0, s20, v0\n v_xor_b32 v8, s20, v8\n v_xor_b32 v2, s21, v2\n v_xor_b32 v10, s21, v10\n" \
                        "v_mov_b32 v1, v6\n v_mov_b32 v9, v14\n v_mov_b32 v3, v4\n v_mov_b32 v11, v12\n"
PATTERN_KERNEL(k_philox_round_2streams, PHILOX_ROUND_2X)


// --- the complete inner-loop body of the path kernel, hand-scheduled: one Philox block (hoisted:
// 17 mads + 34 xors) + Box-Muller (4 cvt, 4 fma, 8 transcendental, 2+4 mul) + 4 OU steps (4 fmac + 4 add).
// v0=A2 v1=Q v2=R v3=L0 v4=T v5=y v6=S ; s20..: keys ; s22=M0 s23=M1 ; s26 = j ; s27 = e
#define FULL_BLOCK \
    "s_mul_hi_u32 s24, s26, s22\n s_mul_i32 s25, s26, s22\n s_add_u32 s26, s26, 1\n" \
    "v_xor_b32 v8, s24, v0\n" \
    "v_mad_u64_u32 v[10:11], s[30:31], v8, s23, 0\n" \
    "v_xor_b32 v9, s25, v2\n v_xor_b32 v8, v11, v1\n" \
    "v_mad_u64_u32 v[12:13], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[14:15], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v15, v10\n v_xor_b32 v9, v13, v4\n v_xor_b32 v8, s20, v8\n" \
    "v_mad_u64_u32 v[16:17], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[18:19], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v19, v14\n v_xor_b32 v9, v17, v12\n v_xor_b32 v8, s20, v8\n v_xor_b32 v9, s21, v9\n" \
    "v_mad_u64_u32 v[12:13], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[14:15], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v15, v18\n v_xor_b32 v9, v13, v16\n v_xor_b32 v8, s21, v8\n v_xor_b32 v9, s20, v9\n" \
    "v_mad_u64_u32 v[16:17], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[18:19], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v19, v14\n v_xor_b32 v9, v17, v12\n v_xor_b32 v8, s20, v8\n v_xor_b32 v9, s21, v9\n" \
    "v_mad_u64_u32 v[12:13], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[14:15], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v15, v18\n v_xor_b32 v9, v13, v16\n v_xor_b32 v8, s21, v8\n v_xor_b32 v9, s20, v9\n" \
    "v_mad_u64_u32 v[16:17], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[18:19], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v19, v14\n v_xor_b32 v9, v17, v12\n v_xor_b32 v8, s20, v8\n v_xor_b32 v9, s21, v9\n" \
    "v_mad_u64_u32 v[12:13], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[14:15], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v15, v18\n v_xor_b32 v9, v13, v16\n v_xor_b32 v8, s21, v8\n v_xor_b32 v9, s20, v9\n" \
    "v_mad_u64_u32 v[16:17], s[30:31], v8, s22, 0\n v_mad_u64_u32 v[18:19], s[30:31], v9, s23, 0\n" \
    "v_xor_b32 v8, v19, v14\n v_xor_b32 v9, v17, v12\n v_xor_b32 v8, s20, v8\n v_xor_b32 v9, s21, v9\n" \
    "v_cvt_f32_u32 v8, v8\n v_cvt_f32_u32 v9, v9\n v_cvt_f32_u32 v10, v18\n v_cvt_f32_u32 v11, v16\n" \
    "v_fma_f32 v8, v8, s28, s28\n v_fma_f32 v9, v9, s28, s28\n v_fma_f32 v10, v10, s28, s28\n v_fma_f32 v11, v11, s28, s28\n" \
    "v_log_f32 v8, v8\n v_log_f32 v9, v9\n v_sin_f32 v12, v10\n v_cos_f32 v10, v10\n" \
    "v_mul_f32 v8, 0xbfb17218, v8\n v_mul_f32 v9, 0xbfb17218, v9\n v_sin_f32 v13, v11\n v_cos_f32 v11, v11\n" \
    "v_sqrt_f32 v8, v8\n v_sqrt_f32 v9, v9\n" \
    "v_mul_f32 v12, v12, v8\n v_mul_f32 v10, v10, v8\n v_mul_f32 v13, v13, v9\n v_mul_f32 v11, v11, v9\n" \
    "v_fmac_f32 v12, s27, v5\n v_add_f32 v6, v6, v12\n v_fmac_f32 v10, s27, v12\n v_add_f32 v6, v6, v10\n" \
    "v_fmac_f32 v13, s27, v10\n v_add_f32 v6, v6, v13\n v_fmac_f32 v11, s27, v13\n v_add_f32 v6, v6, v11\n v_mov_b32 v5, v11\n"

#define PATTERN_KERNEL1(NAME, GROUP)                                                       \
    __global__ void NAME(unsigned* out, int iters, unsigned seed)                         \
    {                                                                                      \
        asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %0\n v_mov_b32 v4, %0\n" \
                     "v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n s_mov_b32 s20, %1\n s_mov_b32 s21, %1\n s_mov_b32 s22, 0xD2511F53\n s_mov_b32 s23, 0xCD9E8D57\n" \
                     "s_mov_b32 s26, 0\n s_mov_b32 s27, 0x3f7d73c0\n s_mov_b32 s28, 0x2f800000\n" \
                     :: "v"(threadIdx.x * 2654435761u + seed), "s"(seed * 77u + 12345u) : CLOBBERS, "s28");  \
        for (int it = 0; it < iters; it++) { asm volatile(R16(GROUP) ::: CLOBBERS, "s28"); }     \
        unsigned r;                                                                        \
        asm volatile("v_xor_b32 %0, v5, v6" : "=v"(r) :: CLOBBERS);                         \
        if (r == 0x12345679u) out[0] = r;                                                  \
    }
PATTERN_KERNEL1(k_full_block, FULL_BLOCK)
// same body, ONE block per loop trip (small loop: 82 instructions) -- code-size sensitivity
#define PATTERN_KERNEL1S(NAME, GROUP)                                                      \
    __global__ void NAME(unsigned* out, int iters, unsigned seed)                         \
    {                                                                                      \
        asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %0\n v_mov_b32 v4, %0\n" \
                     "v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n s_mov_b32 s20, %1\n s_mov_b32 s21, %1\n s_mov_b32 s22, 0xD2511F53\n s_mov_b32 s23, 0xCD9E8D57\n" \
                     "s_mov_b32 s26, 0\n s_mov_b32 s27, 0x3f7d73c0\n s_mov_b32 s28, 0x2f800000\n" \
                     :: "v"(threadIdx.x * 2654435761u + seed), "s"(seed * 77u + 12345u) : CLOBBERS, "s28");  \
        for (int it = 0; it < iters * 16; it++) { asm volatile(GROUP ::: CLOBBERS, "s28"); }     \
        unsigned r;                                                                        \
        asm volatile("v_xor_b32 %0, v5, v6" : "=v"(r) :: CLOBBERS);                         \
        if (r == 0x12345679u) out[0] = r;                                                  \
    }
PATTERN_KERNEL1S(k_full_block_small, FULL_BLOCK)
// Box-Muller + steps only (inputs recycled with 4 cheap xors): 4 cvt, 4 fma, 8 trans, 6 mul, 4 fmac, 4 add, 1 mov + 4 xor
#define BM_BLOCK \
    "v_xor_b32 v8, 0x9E3779B9, v8\n v_xor_b32 v9, v8, v9\n v_xor_b32 v18, 0xBB67AE85, v18\n v_xor_b32 v16, v18, v16\n" \
    "v_cvt_f32_u32 v8, v8\n v_cvt_f32_u32 v9, v9\n v_cvt_f32_u32 v10, v18\n v_cvt_f32_u32 v11, v16\n" \
    "v_fma_f32 v8, v8, s28, s28\n v_fma_f32 v9, v9, s28, s28\n v_fma_f32 v10, v10, s28, s28\n v_fma_f32 v11, v11, s28, s28\n" \
    "v_log_f32 v8, v8\n v_log_f32 v9, v9\n v_sin_f32 v12, v10\n v_cos_f32 v10, v10\n" \
    "v_mul_f32 v8, 0xbfb17218, v8\n v_mul_f32 v9, 0xbfb17218, v9\n v_sin_f32 v13, v11\n v_cos_f32 v11, v11\n" \
    "v_sqrt_f32 v8, v8\n v_sqrt_f32 v9, v9\n" \
    "v_mul_f32 v12, v12, v8\n v_mul_f32 v10, v10, v8\n v_mul_f32 v13, v13, v9\n v_mul_f32 v11, v11, v9\n" \
    "v_fmac_f32 v12, s27, v5\n v_add_f32 v6, v6, v12\n v_fmac_f32 v10, s27, v12\n v_add_f32 v6, v6, v10\n" \
    "v_fmac_f32 v13, s27, v10\n v_add_f32 v6, v6, v13\n v_fmac_f32 v11, s27, v13\n v_add_f32 v6, v6, v11\n v_mov_b32 v5, v11\n"
PATTERN_KERNEL1(k_bm_block, BM_BLOCK)
// 8 transcendentals interleaved 1:1 with 8 full-rate ops vs grouped
PATTERN_KERNEL1(k_trans_interleaved, "v_log_f32 v8, v8\n v_xor_b32 v0, v0, v1\n v_sin_f32 v9, v9\n v_xor_b32 v2, v2, v1\n v_sqrt_f32 v10, v10\n v_xor_b32 v3, v3, v1\n v_cos_f32 v11, v11\n v_xor_b32 v4, v4, v1\n")
PATTERN_KERNEL1(k_trans_grouped, "v_log_f32 v8, v8\n v_sin_f32 v9, v9\n v_sqrt_f32 v10, v10\n v_cos_f32 v11, v11\n v_xor_b32 v0, v0, v1\n v_xor_b32 v2, v2, v1\n v_xor_b32 v3, v3, v1\n v_xor_b32 v4, v4, v1\n")
PATTERN_KERNEL1(k_cvt_fma, "v_cvt_f32_u32 v8, v0\n v_cvt_f32_u32 v9, v1\n v_cvt_f32_u32 v10, v2\n v_cvt_f32_u32 v11, v3\n v_fma_f32 v8, v8, s28, s28\n v_fma_f32 v9, v9, s28, s28\n v_fma_f32 v10, v10, s28, s28\n v_fma_f32 v11, v11, s28, s28\n")
// producer -> dependent consumer pairs, 4 independent chains per group
PATTERN_KERNEL1(k_log_then_mul_dep, "v_log_f32 v8, v8\n v_mul_f32 v8, 0x3f800001, v8\n v_log_f32 v9, v9\n v_mul_f32 v9, 0x3f800001, v9\n v_log_f32 v10, v10\n v_mul_f32 v10, 0x3f800001, v10\n v_log_f32 v11, v11\n v_mul_f32 v11, 0x3f800001, v11\n")
PATTERN_KERNEL1(k_log_then_mul_indep, "v_log_f32 v8, v8\n v_mul_f32 v12, 0x3f800001, v12\n v_log_f32 v9, v9\n v_mul_f32 v13, 0x3f800001, v13\n v_log_f32 v10, v10\n v_mul_f32 v14, 0x3f800001, v14\n v_log_f32 v11, v11\n v_mul_f32 v15, 0x3f800001, v15\n")
PATTERN_KERNEL1(k_log_sqrt_dep, "v_log_f32 v8, v8\n v_sqrt_f32 v8, v8\n v_log_f32 v9, v9\n v_sqrt_f32 v9, v9\n v_log_f32 v10, v10\n v_sqrt_f32 v10, v10\n v_log_f32 v11, v11\n v_sqrt_f32 v11, v11\n")
PATTERN_KERNEL1(k_cvt_then_fma_dep, "v_cvt_f32_u32 v8, v8\n v_fma_f32 v8, v8, s28, s28\n v_cvt_f32_u32 v9, v9\n v_fma_f32 v9, v9, s28, s28\n v_cvt_f32_u32 v10, v10\n v_fma_f32 v10, v10, s28, s28\n v_cvt_f32_u32 v11, v11\n v_fma_f32 v11, v11, s28, s28\n")
PATTERN_KERNEL1(k_cvt_only, "v_cvt_f32_u32 v8, v0\n v_cvt_f32_u32 v9, v1\n v_cvt_f32_u32 v10, v2\n v_cvt_f32_u32 v11, v3\n v_cvt_f32_u32 v12, v0\n v_cvt_f32_u32 v13, v1\n v_cvt_f32_u32 v14, v2\n v_cvt_f32_u32 v15, v3\n")
PATTERN_KERNEL1(k_mad_then_xor_dep, "v_mad_u64_u32 v[8:9], s[30:31], v9, s22, 0\n v_xor_b32 v9, v9, v8\n v_mad_u64_u32 v[10:11], s[30:31], v11, s23, 0\n v_xor_b32 v11, v11, v10\n v_mad_u64_u32 v[12:13], s[30:31], v13, s22, 0\n v_xor_b32 v13, v13, v12\n v_mad_u64_u32 v[14:15], s[30:31], v15, s23, 0\n v_xor_b32 v15, v15, v14\n")
PATTERN_KERNEL1(k_mad_xor_indep, "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_xor_b32 v16, v16, v1\n v_mad_u64_u32 v[10:11], s[30:31], v1, s23, 0\n v_xor_b32 v17, v17, v1\n v_mad_u64_u32 v[12:13], s[30:31], v2, s22, 0\n v_xor_b32 v18, v18, v1\n v_mad_u64_u32 v[14:15], s[30:31], v3, s23, 0\n v_xor_b32 v19, v19, v1\n")
PATTERN_KERNEL1(k_mad_2xor_indep, "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_xor_b32 v16, v16, v1\n v_xor_b32 v17, v17, v1\n v_mad_u64_u32 v[10:11], s[30:31], v1, s23, 0\n v_xor_b32 v18, v18, v1\n v_xor_b32 v19, v19, v1\n v_xor_b32 v16, v16, v2\n v_xor_b32 v17, v17, v2\n")
// two chained Philox rounds without moves (register sets alternate): c0=v0 c1=v1 c2=v2 c3=v3
#define ROUND2_MAD \
    "v_mad_u64_u32 v[4:5], s[30:31], v0, s22, 0\n v_mad_u64_u32 v[6:7], s[30:31], v2, s23, 0\n" \
    "v_xor_b32 v0, v7, v1\n v_xor_b32 v2, v5, v3\n v_xor_b32 v0, s20, v0\n v_xor_b32 v2, s21, v2\n" \
    "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_mad_u64_u32 v[10:11], s[30:31], v2, s23, 0\n" \
    "v_xor_b32 v0, v11, v6\n v_xor_b32 v2, v9, v4\n v_xor_b32 v0, s21, v0\n v_xor_b32 v2, s20, v2\n v_mov_b32 v1, v10\n v_mov_b32 v3, v8\n"
PATTERN_KERNEL1(k_round2_mad, ROUND2_MAD)
#define ROUND2_MULHILO \
    "v_mul_hi_u32 v5, v0, s22\n v_mul_hi_u32 v7, v2, s23\n v_mul_lo_u32 v4, v0, s22\n v_mul_lo_u32 v6, v2, s23\n" \
    "v_xor_b32 v0, v7, v1\n v_xor_b32 v2, v5, v3\n v_xor_b32 v0, s20, v0\n v_xor_b32 v2, s21, v2\n" \
    "v_mul_hi_u32 v9, v0, s22\n v_mul_hi_u32 v11, v2, s23\n v_mul_lo_u32 v8, v0, s22\n v_mul_lo_u32 v10, v2, s23\n" \
    "v_xor_b32 v0, v11, v6\n v_xor_b32 v2, v9, v4\n v_xor_b32 v0, s21, v0\n v_xor_b32 v2, s20, v2\n v_mov_b32 v1, v10\n v_mov_b32 v3, v8\n"
PATTERN_KERNEL1(k_round2_mulhilo, ROUND2_MULHILO)
// key xor applied to the low word early (off the critical path), one xor after the mads
#define ROUND2_MAD_EARLYKEY \
    "v_mad_u64_u32 v[4:5], s[30:31], v0, s22, 0\n v_mad_u64_u32 v[6:7], s[30:31], v2, s23, 0\n" \
    "v_xor_b32 v1, s20, v1\n v_xor_b32 v3, s21, v3\n v_xor_b32 v0, v7, v1\n v_xor_b32 v2, v5, v3\n" \
    "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_mad_u64_u32 v[10:11], s[30:31], v2, s23, 0\n" \
    "v_xor_b32 v6, s21, v6\n v_xor_b32 v4, s20, v4\n v_xor_b32 v0, v11, v6\n v_xor_b32 v2, v9, v4\n v_mov_b32 v1, v10\n v_mov_b32 v3, v8\n"
PATTERN_KERNEL1(k_round2_mad_earlykey, ROUND2_MAD_EARLYKEY)
// mads separated: mad, xor-pair, mad, xor-pair (worst case interleave)
#define ROUND2_MAD_SPLIT \
    "v_mad_u64_u32 v[4:5], s[30:31], v0, s22, 0\n v_xor_b32 v12, s20, v1\n v_mad_u64_u32 v[6:7], s[30:31], v2, s23, 0\n v_xor_b32 v13, s21, v3\n" \
    "v_xor_b32 v0, v7, v12\n v_xor_b32 v2, v5, v13\n" \
    "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_xor_b32 v12, s21, v6\n v_mad_u64_u32 v[10:11], s[30:31], v2, s23, 0\n v_xor_b32 v13, s20, v4\n" \
    "v_xor_b32 v0, v11, v12\n v_xor_b32 v2, v9, v13\n v_mov_b32 v1, v10\n v_mov_b32 v3, v8\n"
PATTERN_KERNEL1(k_round2_mad_split, ROUND2_MAD_SPLIT)
#define T1 "v_cvt_f32_u32 v8, v0\n v_cvt_f32_u32 v9, v1\n v_cvt_f32_u32 v10, v2\n v_cvt_f32_u32 v11, v3\n v_fma_f32 v8, v8, s28, s28\n v_fma_f32 v9, v9, s28, s28\n v_fma_f32 v10, v10, s28, s28\n v_fma_f32 v11, v11, s28, s28\n"
#define T2 "v_log_f32 v8, v8\n v_log_f32 v9, v9\n v_sin_f32 v12, v10\n v_cos_f32 v10, v10\n v_mul_f32 v8, 0xbfb17218, v8\n v_mul_f32 v9, 0xbfb17218, v9\n v_sin_f32 v13, v11\n v_cos_f32 v11, v11\n v_sqrt_f32 v8, v8\n v_sqrt_f32 v9, v9\n"
#define T3 "v_mul_f32 v12, v12, v8\n v_mul_f32 v10, v10, v8\n v_mul_f32 v13, v13, v9\n v_mul_f32 v11, v11, v9\n" \
    "v_fmac_f32 v12, s27, v5\n v_add_f32 v6, v6, v12\n v_fmac_f32 v10, s27, v12\n v_add_f32 v6, v6, v10\n" \
    "v_fmac_f32 v13, s27, v10\n v_add_f32 v6, v6, v13\n v_fmac_f32 v11, s27, v13\n v_add_f32 v6, v6, v11\n v_mov_b32 v5, v11\n"
PATTERN_KERNEL1(k_T1, T1)
PATTERN_KERNEL1(k_T2, T2)
PATTERN_KERNEL1(k_T3, T3)
PATTERN_KERNEL1(k_T12, T1 T2)
PATTERN_KERNEL1(k_T23, T2 T3)
PATTERN_KERNEL1(k_T123, T1 T2 T3)
PATTERN_KERNEL1(k_trans8, "v_log_f32 v8, v8\n v_log_f32 v9, v9\n v_sin_f32 v12, v10\n v_cos_f32 v10, v10\n v_sin_f32 v13, v11\n v_cos_f32 v11, v11\n v_sqrt_f32 v8, v8\n v_sqrt_f32 v9, v9\n")
PATTERN_KERNEL1(k_mad_trans_alt, "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_log_f32 v16, v16\n v_mad_u64_u32 v[10:11], s[30:31], v1, s23, 0\n v_sin_f32 v17, v17\n v_mad_u64_u32 v[12:13], s[30:31], v2, s22, 0\n v_sqrt_f32 v18, v18\n v_mad_u64_u32 v[14:15], s[30:31], v3, s23, 0\n v_cos_f32 v19, v19\n")
PATTERN_KERNEL1(k_mad2_trans2, "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_mad_u64_u32 v[10:11], s[30:31], v1, s23, 0\n v_log_f32 v16, v16\n v_sin_f32 v17, v17\n v_mad_u64_u32 v[12:13], s[30:31], v2, s22, 0\n v_mad_u64_u32 v[14:15], s[30:31], v3, s23, 0\n v_sqrt_f32 v18, v18\n v_cos_f32 v19, v19\n")
PATTERN_KERNEL1(k_cvt_trans_alt, "v_cvt_f32_u32 v8, v0\n v_log_f32 v16, v16\n v_cvt_f32_u32 v9, v1\n v_sin_f32 v17, v17\n v_cvt_f32_u32 v10, v2\n v_sqrt_f32 v18, v18\n v_cvt_f32_u32 v11, v3\n v_cos_f32 v19, v19\n")
PATTERN_KERNEL1(k_mad_cvt_alt, "v_mad_u64_u32 v[8:9], s[30:31], v0, s22, 0\n v_cvt_f32_u32 v16, v0\n v_mad_u64_u32 v[10:11], s[30:31], v1, s23, 0\n v_cvt_f32_u32 v17, v1\n v_mad_u64_u32 v[12:13], s[30:31], v2, s22, 0\n v_cvt_f32_u32 v18, v2\n v_mad_u64_u32 v[14:15], s[30:31], v3, s23, 0\n v_cvt_f32_u32 v19, v3\n")
PATTERN_KERNEL1(k_xor8_indep, "v_xor_b32 v8, v0, v1\n v_xor_b32 v9, v1, v2\n v_xor_b32 v10, v2, v3\n v_xor_b32 v11, v3, v4\n v_xor_b32 v12, v0, v2\n v_xor_b32 v13, v1, v3\n v_xor_b32 v14, v2, v4\n v_xor_b32 v15, v3, v0\n")
PATTERN_KERNEL1(k_fma_2sgpr, "v_fma_f32 v8, v8, s28, s28\n v_fma_f32 v9, v9, s28, s28\n v_fma_f32 v10, v10, s28, s28\n v_fma_f32 v11, v11, s28, s28\n v_fma_f32 v12, v12, s28, s28\n v_fma_f32 v13, v13, s28, s28\n v_fma_f32 v14, v14, s28, s28\n v_fma_f32 v15, v15, s28, s28\n")
PATTERN_KERNEL1(k_fmac_sgpr_add, "v_fmac_f32 v12, s27, v5\n v_add_f32 v6, v6, v12\n v_fmac_f32 v10, s27, v12\n v_add_f32 v6, v6, v10\n v_fmac_f32 v13, s27, v10\n v_add_f32 v6, v6, v13\n v_fmac_f32 v11, s27, v13\n v_add_f32 v6, v6, v11\n")
typedef void (*kern_t)(unsigned*, int, unsigned);
struct Entry { const char* name; kern_t fn; int instr_per_group; };

int main(int argc, char** argv)
{
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    const int block = 256, grid = cus * wps, iters = 200;
    Entry tab[] = {
        {"xor same-bank srcs", k_xor_samebank, 8}, {"xor diff-bank srcs", k_xor_diffbank, 8}, {"xor sgpr src", k_xor_sgpr, 8},
        {"xor literal src", k_xor_literal, 8}, {"xor dep chain x1", k_xor_dep1, 8}, {"xor dep chain x2", k_xor_dep2, 8},
        {"mad_u64 indep", k_mad_indep, 8}, {"mad_u64 dep x2", k_mad_dep2, 8}, {"mul_hi+mul_lo", k_mulhi_mullo, 8},
        {"philox round sgpr key (8 instr)", k_philox_round_sgprkey, 8}, {"philox round vgpr key", k_philox_round_vgprkey, 8},
        {"philox round literal key", k_philox_round_literalkey, 8}, {"philox round 2 streams (16)", k_philox_round_2streams, 16},
        {"FULL block: philox+BM+4 steps (82)", k_full_block, 82},
        {"FULL block, 1 per loop trip", k_full_block_small, 82}, {"BM+steps block (39)", k_bm_block, 39},
        {"4 trans + 4 xor interleaved", k_trans_interleaved, 8}, {"4 trans + 4 xor grouped", k_trans_grouped, 8},
        {"log->mul dependent x4", k_log_then_mul_dep, 8}, {"log, mul independent x4", k_log_then_mul_indep, 8},
        {"log->sqrt dependent x4", k_log_sqrt_dep, 8}, {"cvt->fma dependent x4", k_cvt_then_fma_dep, 8}, {"8 cvt", k_cvt_only, 8},
        {"mad->xor dependent x4", k_mad_then_xor_dep, 8}, {"mad, xor independent x4", k_mad_xor_indep, 8},
        {"2 mad + 6 xor independent", k_mad_2xor_indep, 8},
        {"2 rounds mad (14)", k_round2_mad, 14}, {"2 rounds mul_hi/lo (18)", k_round2_mulhilo, 18},
        {"2 rounds mad, key early (14)", k_round2_mad_earlykey, 14}, {"2 rounds mad split (14)", k_round2_mad_split, 14},
        {"T1: 4cvt+4fma (8)", k_T1, 8}, {"T2: 8 trans + 2 mul (10)", k_T2, 10}, {"T3: 4 mul + 8 step + mov (13)", k_T3, 13},
        {"T1+T2 (18)", k_T12, 18}, {"T2+T3 (23)", k_T23, 23}, {"T1+T2+T3 (31)", k_T123, 31}, {"8 trans mixed kinds", k_trans8, 8},
        {"mad,trans alternating (8)", k_mad_trans_alt, 8}, {"mad mad trans trans (8)", k_mad2_trans2, 8}, {"cvt,trans alternating", k_cvt_trans_alt, 8},
        {"mad,cvt alternating", k_mad_cvt_alt, 8}, {"8 xor independent", k_xor8_indep, 8},
        {"4 cvt + 4 fma(s,s)", k_cvt_fma, 8}, {"8 fma v,s,s", k_fma_2sgpr, 8}, {"4 fmac(s) + 4 add chain", k_fmac_sgpr_add, 8},
    };
    printf("device %s CUs=%d  %d waves/SIMD, cycles at nominal 2.4 GHz\n", prop.gcnArchName, cus, wps);
    printf("%-34s %9s %14s %14s\n", "pattern", "ms(min)", "cyc/instr", "cyc/group");
    for (const Entry& en : tab) {
        std::vector<float> ms;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(block), 0, 0, d_out, iters, (unsigned)rep);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (rep > 0) ms.push_back(t);
        }
        const double tmin = *std::min_element(ms.begin(), ms.end());
        const double groups = (double)grid * (block / 64) * (double)iters * 16;
        const double cyc_group = tmin * 1e-3 * 2.4e9 * (cus * 4.0) / groups;
        printf("%-34s %9.3f %14.2f %14.2f\n", en.name, tmin, cyc_group / en.instr_per_group, cyc_group);
    }
    return 0;
}
