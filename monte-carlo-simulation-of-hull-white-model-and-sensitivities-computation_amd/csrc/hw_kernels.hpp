// hw_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// One thread walks one antithetic pair (ZBC, curve) or one path (Vega).  Nothing is read from
// HBM inside the time loop: the Philox4x32-10 counter is built in registers from
// (seed, global path id, step block), and the only global traffic of a launch is one small
// row of double partial sums per workgroup.  The kernels are VALU-issue-bound (integer
// multiplies of Philox + the four transcendentals of Box-Muller), not HBM-bound; see DESIGN.md.
//
// Reference kernels these replace (paths relative to the reference root):
//   zbc_superposed_kernel / zbc_reference_kernel : simulate_ZBC_control_variate, include/common.cuh:286-409
//   zcb_curve_kernel                             : simulate_zcb, include/market_data.cuh:25-79
//   vega_pathwise_kernel                         : simulate_sensitivity, src/3_sensitivity_analysis.cu:22-96
//   sample_paths_kernel                          : simulate_paths_show, include/market_data.cuh:136-160
//   (init_rng, include/common.cuh:277-280, has no counterpart: there is no RNG state to initialise)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace hwk {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kMaxWaves = 16;      // 1024-thread workgroup
constexpr int kMaxScen = 3;

// ------------------------------------------------------------------------------------------
// Philox4x32-10, rocRAND counter layout: counter = {block j, 0, path lo, path hi}, key = seed.
// The ten round keys depend only on the seed (wave-uniform).  Because counter words 0/1 are
// wave-uniform and words 2/3 are constant along a path, round 1 and half of rounds 2-3 are
// loop-invariant per lane: philox_path() hoists them, philox_block() spends 17 wide multiplies
// (16 in philox_block_prio<UHI = true>) and 19 bitwise ops (v_bitop3_b32 = 3-input xor) per block
// instead of rocRAND's 20 and 40.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u, kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;

struct PhiloxKey {
    uint32_t k0[10], k1[10];
    uint32_t s_k0_2;      // k0[2] again: stays in an SGPR when philox_key_to_vgprs() moves k0[2..9], k1[2..9]
};

__device__ __forceinline__ PhiloxKey philox_key(uint64_t seed)
{
    PhiloxKey k;
    uint32_t a = (uint32_t)seed, b = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        k.k0[r] = a;
        k.k1[r] = b;
        a += kW0;
        b += kW1;
    }
    k.s_k0_2 = k.k0[2];
    return k;
}

// a ^ b ^ c in ONE instruction: gfx950's v_bitop3_b32 (any three-input bitwise function, truth table 0x96 =
// parity) issues at the rate of v_xor_b32 (tools/valu_rates.hip: 2.45 vs 2.47 cycles), so the key-mixing of a
// Philox round costs 2 instructions instead of 4: 34 -> 19 bitwise ops per block.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Round keys 3..10 are consumed by every block of the loop: they are copied to vector registers once per
// thread (an SGPR operand on back-to-back v_bitop3 costs 4.1 cycles instead of 2.45).
__device__ __forceinline__ void philox_key_to_vgprs(PhiloxKey& k)
{
#pragma unroll
    for (int r = 2; r < 10; r++) {
        uint32_t a, b;
        asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "s"(k.k0[r]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "s"(k.k1[r]));
        k.k0[r] = a;
        k.k1[r] = b;
    }
}

__device__ __forceinline__ void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo)
{
    const uint64_t p = (uint64_t)a * (uint64_t)b;   // one v_mad_u64_u32
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

struct PhiloxPath {       // per-lane loop invariants of one subsequence
    uint32_t A2, Q, R, L0, T;
};

__device__ __forceinline__ PhiloxPath philox_path(const PhiloxKey& k, uint64_t pid)
{
    const uint32_t plo = (uint32_t)pid, phi = (uint32_t)(pid >> 32);
    uint32_t hi1, lo1, H0, L0;
    mulhilo(kM1, plo, hi1, lo1);                    // round 1, word-2 product
    const uint32_t c0 = hi1 ^ k.k0[0];              // (counter word 1 is 0)
    mulhilo(kM0, c0, H0, L0);                       // round 2, word-0 product
    PhiloxPath pp;
    pp.A2 = phi ^ k.k1[0];
    pp.Q = lo1 ^ k.k0[1];
    pp.R = H0 ^ k.k1[1];
    pp.L0 = L0;
    pp.T = L0 ^ k.k1[2];
    return pp;
}

// Block j (wave-uniform) of the path: the four 32-bit outputs of rocrand4().
__device__ __forceinline__ void philox_block(const PhiloxKey& k, const PhiloxPath& pp, uint32_t j,
                                             uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(kM0, j, hi0, lo0);                      // round 1 (scalar: j is uniform)
    uint32_t c2 = pp.A2 ^ hi0;
    uint32_t c3 = lo0;
    mulhilo(kM1, c2, hi1, lo1);                     // round 2
    uint32_t c0 = hi1 ^ pp.Q;
    uint32_t c1 = lo1;
    c2 = pp.R ^ c3;
    mulhilo(kM0, c0, hi0, lo0);                     // round 3
    mulhilo(kM1, c2, hi1, lo1);
    c0 = xor3(hi1, c1, k.k0[2]);
    c1 = lo1;
    c2 = hi0 ^ pp.T;
    c3 = lo0;
#pragma unroll
    for (int r = 3; r < 10; r++) {                  // rounds 4..10
        mulhilo(kM0, c0, hi0, lo0);
        mulhilo(kM1, c2, hi1, lo1);
        c0 = xor3(hi1, c1, k.k0[r]);
        c1 = lo1;
        c2 = xor3(hi0, c3, k.k1[r]);
        c3 = lo0;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// Reference (unhoisted) form, used by the debug kernel to cross-check the hoisting.
__device__ __forceinline__ void philox_block_plain(uint64_t seed, uint64_t pid, uint64_t j, uint32_t* o)
{
    uint32_t c0 = (uint32_t)j, c1 = (uint32_t)(j >> 32), c2 = (uint32_t)pid, c3 = (uint32_t)(pid >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(kM0, c0, hi0, lo0);
        mulhilo(kM1, c2, hi1, lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += kW0; k1 += kW1;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// Box-Muller as rocrand_normal.h:53-68, on the hardware transcendentals:
//   u = 2^-32 + x 2^-32 in (0,1];  s = sqrt(-2 ln u) = v_sqrt(v_log(u) * -2 ln 2)
//   angle = 2 pi (2^-32 + y 2^-32); v_sin_f32 / v_cos_f32 take the angle in revolutions.
__device__ __forceinline__ void box_muller(uint32_t x, uint32_t y, float& n0, float& n1)
{
    constexpr float kInv = 2.3283064e-10f;          // ROCRAND_2POW32_INV
    const float u = __builtin_fmaf((float)x, kInv, kInv);
    const float w = __builtin_fmaf((float)y, kInv, kInv);
    const float s = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u) * -1.3862943611198906f);
    n0 = __builtin_amdgcn_sinf(w) * s;
    n1 = __builtin_amdgcn_cosf(w) * s;
}

__device__ __forceinline__ void normal4(const PhiloxKey& k, const PhiloxPath& pp, uint32_t j, float (&g)[4])
{
    uint32_t w0, w1, w2, w3;
    philox_block(k, pp, j, w0, w1, w2, w3);
    box_muller(w0, w1, g[0], g[1]);
    box_muller(w2, w3, g[2], g[3]);
}

// ------------------------------------------------------------------------------------------
// Issue-priority choreography of the hot loop (measured on MI355X, tools/labs/loop_lab.hip family L5:
// 312 -> 272 cycles per Philox block).  A gfx950 SIMD issues single-rate VALU ops of two waves side by
// side, but v_mad_u64_u32, v_cvt and the transcendentals run alone.  With all waves at equal priority
// the exclusive ops of one wave keep splitting the pairs of the others; raising the wave's priority
// (s_setprio 1) exactly while it issues its exclusive ops -- the two multiplies of each Philox round
// and the whole Box-Muller/step tail -- and dropping it for the xors lets the arbiter drain exclusive
// work first and pair the single-rate rest.  s_setprio is not a scheduling barrier for pure
// arithmetic, so the phase boundaries are pinned with empty volatile asm statements that carry the
// live values through (no instructions are emitted for them).
// ------------------------------------------------------------------------------------------
#define HWK_SETPRIO(n)                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_setprio(n);        \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

// the toggles INSIDE the Philox rounds (block-level phases only, or priorities 0/2/3 instead of 0/1, measured slower:
// DESIGN.md section 5)
#define HWK_SETPRIO_R(n) HWK_SETPRIO(n)

__device__ __forceinline__ void pin2(uint32_t& a, uint32_t& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void pin4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// Same words as philox_block().  Entered AND left at priority 1: the caller's Box-Muller + steps run
// there, and so do the three multiplies of rounds 2-3 (measured: 272 -> 265 cycles per block against
// dropping the priority at block entry); the xors run at priority 0.
//
// UHI = true: the HIGH word of the path id is the same in every lane of the wave (a2s = that word ^ k1[0], in
// an SGPR) -- true for every wave that does not straddle a multiple of 2^32 paths.  Then word 2 after round 1,
// hi(M0 j) ^ phi ^ k1[0], is wave-uniform as well, and the M1 multiply of round 2 runs on the scalar unit:
// 16 vector multiplies and 18 bitwise ops per block instead of 17 and 19.
template <bool UHI, bool RAISE = true, int BASE = 0>
__device__ __forceinline__ void philox_block_prio(const PhiloxKey& k, const PhiloxPath& pp, uint32_t a2s, uint32_t j,
                                                  uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(kM0, j, hi0, lo0);                      // round 1 (scalar)
    uint32_t c0, c1, c2, c3;
    if constexpr (UHI) {
        uint32_t shi, slo;
        mulhilo(kM1, a2s ^ hi0, shi, slo);          // round 2, word-2 product (scalar)
        c0 = shi ^ pp.Q;
        c2 = pp.R ^ lo0;
        mulhilo(kM0, c0, hi0, lo0);                 // round 3
        mulhilo(kM1, c2, hi1, lo1);
        pin2(hi0, hi1);
        HWK_SETPRIO(BASE + 0);
        pin2(hi0, hi1);
        c0 = hi1 ^ (slo ^ k.s_k0_2);
    } else {
        c2 = pp.A2 ^ hi0;
        c3 = lo0;
        mulhilo(kM1, c2, hi1, lo1);                 // round 2
        c0 = hi1 ^ pp.Q;
        c1 = lo1;
        c2 = pp.R ^ c3;
        mulhilo(kM0, c0, hi0, lo0);                 // round 3
        mulhilo(kM1, c2, hi1, lo1);
        pin2(hi0, hi1);
        HWK_SETPRIO(BASE + 0);
        pin2(hi0, hi1);
        c0 = xor3(hi1, c1, k.k0[2]);
    }
    c1 = lo1;
    c2 = hi0 ^ pp.T;
    c3 = lo0;
#pragma unroll
    for (int r = 3; r < 10; r++) {                  // rounds 4..10: multiplies at priority 1, xors at 0
        pin2(c0, c2);
        HWK_SETPRIO_R(BASE + 1);
        pin2(c0, c2);
        mulhilo(kM0, c0, hi0, lo0);
        mulhilo(kM1, c2, hi1, lo1);
        pin2(hi0, hi1);
        HWK_SETPRIO_R(BASE + 0);
        pin2(hi0, hi1);
        c0 = xor3(hi1, c1, k.k0[r]);
        c1 = lo1;
        c2 = xor3(hi0, c3, k.k1[r]);
        c3 = lo0;
    }
    if constexpr (RAISE) {
        pin4(c0, c1, c2, c3);
        HWK_SETPRIO(BASE + 1);
        pin4(c0, c1, c2, c3);
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
__device__ __forceinline__ void philox_block_prio(const PhiloxKey& k, const PhiloxPath& pp, uint32_t j,
                                                  uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
    philox_block_prio<false>(k, pp, 0u, j, o0, o1, o2, o3);
}

// Block j of TWO paths of one thread in lockstep (pairs in flight = 2): the four multiplies of a round back to back at
// priority 1, the four key mixes at priority 0 -- two independent dependency chains per wave and half as many priority
// toggles per pair.  Same words as two philox_block_prio<UHI> calls.  What it buys is half as many waves, i.e. half the
// per-wave work outside the loop: +1.0 % at 2^22 pairs, +1.2 % at 2^26 with 500 steps per path (profiles/r02_ilp_pipeline_probe.txt; see
// effective_ilp() in hw_engine.hip for when the engine uses it).  Needs -fno-slp-vectorize: left alone, the compiler packs the
// two pairs' step arithmetic into v_pk_* instructions and pays 12 v_mov per block to build their operands (-5 %).
template <bool UHI>
__device__ __forceinline__ void philox_block_prio_x2(const PhiloxKey& k, const PhiloxPath (&pp)[2], const uint32_t (&a2s)[2],
                                                     uint32_t j, uint32_t (&o)[2][4])
{
    uint32_t hi0[2], lo0[2], hi1[2], lo1[2], c0[2], c1[2], c2[2], c3[2];
    uint32_t rh, rl;
    mulhilo(kM0, j, rh, rl);                        // round 1 (scalar, shared by both paths)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if constexpr (UHI) {
            uint32_t shi, slo;
            mulhilo(kM1, a2s[q] ^ rh, shi, slo);    // round 2, word-2 product (scalar)
            c0[q] = shi ^ pp[q].Q;
            c2[q] = pp[q].R ^ rl;
            mulhilo(kM0, c0[q], hi0[q], lo0[q]);    // round 3
            mulhilo(kM1, c2[q], hi1[q], lo1[q]);
            c1[q] = slo ^ k.s_k0_2;                 // (scalar) folded into the first key mix below
        } else {
            c2[q] = pp[q].A2 ^ rh;
            c3[q] = rl;
            mulhilo(kM1, c2[q], hi1[q], lo1[q]);    // round 2
            c0[q] = hi1[q] ^ pp[q].Q;
            c1[q] = lo1[q];
            c2[q] = pp[q].R ^ c3[q];
            mulhilo(kM0, c0[q], hi0[q], lo0[q]);    // round 3
            mulhilo(kM1, c2[q], hi1[q], lo1[q]);
        }
    }
    pin4(hi0[0], hi1[0], hi0[1], hi1[1]);
    HWK_SETPRIO(0);
    pin4(hi0[0], hi1[0], hi0[1], hi1[1]);
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if constexpr (UHI) c0[q] = hi1[q] ^ c1[q];
        else c0[q] = xor3(hi1[q], c1[q], k.k0[2]);
        c1[q] = lo1[q];
        c2[q] = hi0[q] ^ pp[q].T;
        c3[q] = lo0[q];
    }
#pragma unroll
    for (int r = 3; r < 10; r++) {                  // rounds 4..10
        pin4(c0[0], c2[0], c0[1], c2[1]);
        HWK_SETPRIO_R(1);
        pin4(c0[0], c2[0], c0[1], c2[1]);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            mulhilo(kM0, c0[q], hi0[q], lo0[q]);
            mulhilo(kM1, c2[q], hi1[q], lo1[q]);
        }
        pin4(hi0[0], hi1[0], hi0[1], hi1[1]);
        HWK_SETPRIO_R(0);
        pin4(hi0[0], hi1[0], hi0[1], hi1[1]);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            c0[q] = xor3(hi1[q], c1[q], k.k0[r]);
            c1[q] = lo1[q];
            c2[q] = xor3(hi0[q], c3[q], k.k1[r]);
            c3[q] = lo0[q];
        }
    }
    pin4(c0[0], c1[0], c2[0], c3[0]);
    pin4(c0[1], c1[1], c2[1], c3[1]);
    HWK_SETPRIO(1);
    pin4(c0[0], c1[0], c2[0], c3[0]);
    pin4(c0[1], c1[1], c2[1], c3[1]);
#pragma unroll
    for (int q = 0; q < 2; q++) { o[q][0] = c0[q]; o[q][1] = c1[q]; o[q][2] = c2[q]; o[q][3] = c3[q]; }
}
#ifndef HWK_ILP2_INTERLEAVE
#define HWK_ILP2_INTERLEAVE 1     // 0: two pairs in flight run their blocks one after the other (round 1's form: never faster)
#endif

// Is the high word of the path id the same in all lanes?  Returns it (xor k1[0]) in an SGPR.
__device__ __forceinline__ bool philox_uniform_hi(const PhiloxPath& pp, uint32_t& a2s)
{
    a2s = __builtin_amdgcn_readfirstlane(pp.A2);
    return __builtin_amdgcn_ballot_w64(pp.A2 != a2s) == 0;
}

// Box-Muller with the constant factor of the radius left out: sqrt(-2 ln u) = kBmScale * sqrt(-log2 u),
// kBmScale = sqrt(2 ln 2).  The hot loops run their linear recursions on n / kBmScale and the host folds
// kBmScale into the epilogue constants (exact algebra; saves the two scaling multiplies per block:
// 270 -> 263 cycles).  v_sqrt_f32 takes the negated logarithm through its source modifier.
constexpr double kBmScale = 1.1774100225154747;
__device__ __forceinline__ void box_muller_unscaled(uint32_t x, uint32_t y, float& n0, float& n1)
{
    constexpr float kInv = 2.3283064e-10f;
    const float u = __builtin_fmaf((float)x, kInv, kInv);
    const float w = __builtin_fmaf((float)y, kInv, kInv);
    const float s = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u));
    n0 = __builtin_amdgcn_sinf(w) * s;
    n1 = __builtin_amdgcn_cosf(w) * s;
}

// The four normals of block j, divided by kBmScale, for the hot loops; exits at priority 1.
template <bool UHI>
__device__ __forceinline__ void normal4_prio(const PhiloxKey& k, const PhiloxPath& pp, uint32_t a2s, uint32_t j, float (&g)[4])
{
    uint32_t w0, w1, w2, w3;
    philox_block_prio<UHI>(k, pp, a2s, j, w0, w1, w2, w3);
    box_muller_unscaled(w0, w1, g[0], g[1]);
    box_muller_unscaled(w2, w3, g[2], g[3]);
}
__device__ __forceinline__ void normal4_prio(const PhiloxKey& k, const PhiloxPath& pp, uint32_t j, float (&g)[4])
{
    normal4_prio<false>(k, pp, 0u, j, g);
}

// ------------------------------------------------------------------------------------------
// small math helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float  hw_exp2(float x)  { return __builtin_amdgcn_exp2f(x); }
// exp2 of a double through the float unit: 2^x = 2^xf (1 + ln2 (x - xf)) with xf = x rounded to float (|x - xf| <=
// 2^-25 |x|: the first-order term is exact to 1e-15).  The result carries v_exp_f32's own error (measured against libm
// over 4 M arguments, tools/labs/exp2_probe.py: max 8.4e-8 relative, sd 2.6e-8, mean signed error -2.3e-9) -- the double engine consumes float normals from the same hardware units (|error| <= 2e-5
// absolute on a normal), so a correctly rounded double exp2 (ocml: ~200 issue cycles a call,
// measured as +5.6 % on the three-scenario kernel at 2^20 pairs) would buy nothing.  |x| < 126 on this path (x ~ O(1)).
__device__ __forceinline__ double hw_exp2(double x)
{
    const float xf = (float)x;
    const double e = (double)__builtin_amdgcn_exp2f(xf);
    return __builtin_fma(e * 0.69314718055994531, x - (double)xf, e);
}
__device__ __forceinline__ float  hw_fma(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double hw_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float  hw_max0(float a)  { return fmaxf(a, 0.0f); }
__device__ __forceinline__ double hw_max0(double a) { return fmax(a, 0.0); }

// Optional per-workgroup timeline (diagnostic; tools/timeline_probe.hip): s_memrealtime (100 MHz) at the entry and
// exit of every workgroup into stamps[2 b], stamps[2 b + 1].  The engine always passes nullptr.
struct WorkgroupStamp {
    unsigned long long t0;
    __device__ __forceinline__ void begin(const unsigned long long* stamps)
    {
        t0 = 0;
        if (stamps != nullptr && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void end(unsigned long long* stamps) const
    {
        if (stamps != nullptr && threadIdx.x == 0) {
            stamps[2 * (size_t)blockIdx.x] = t0;
            stamps[2 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
};

// Optional clock probe (diagnostic; hw_profile_clock*): up to kClkProbes (64) workgroups spread evenly over the grid -- workgroup
// j << shift, j < kClkProbes -- store {s_memtime (shader clock), s_memrealtime (100 MHz)} at their entry and exit into
// clk[4 j .. 4 j + 3].  Workgroups are dispatched in index order, so the probes sample the clock along the whole launch: ONE
// probe in workgroup 0 (rounds 1-3) saw only the first slot of a launch whose grid is larger than the chip, and the clock
// of a 40-100 ms launch is not constant (tools/labs/clock_vs_duration.py).  All conditions are wave-uniform scalars.
constexpr int kClkProbes = 64;
struct ClockProbe {
    __device__ static __forceinline__ bool mine(const unsigned long long* clk, uint32_t shift, uint32_t& j)
    {
        j = blockIdx.x >> shift;
        return clk != nullptr && (j << shift) == blockIdx.x && j < (uint32_t)kClkProbes && threadIdx.x == 0;
    }
    __device__ static __forceinline__ void begin(unsigned long long* clk, uint32_t shift)
    {
        uint32_t j;
        if (mine(clk, shift, j)) {
            clk[4 * j + 0] = __builtin_amdgcn_s_memtime();
            clk[4 * j + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
    __device__ static __forceinline__ void end(unsigned long long* clk, uint32_t shift)
    {
        uint32_t j;
        if (mine(clk, shift, j)) {
            clk[4 * j + 2] = __builtin_amdgcn_s_memtime();
            clk[4 * j + 3] = __builtin_amdgcn_s_memrealtime();
        }
    }
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// float sum over the 64 lanes with DPP row operations (no LDS crossbar traffic): the total lands
// in lane 63.  quad_perm xor-1, xor-2, row_half_mirror, row_mirror give every lane its row-of-16
// sum; row_bcast:15 / row_bcast:31 carry the row sums across the four rows.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += dpp_mov<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_mov<0x140, 0xf>(v);   // row_mirror
    v += dpp_mov<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v += dpp_mov<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

// The same for a double: both halves travel through the DPP network (2 v_mov_b32_dpp + 1 v_add_f64 per step;
// the ds_bpermute form of __shfl_down costs two LDS-crossbar round trips of ~100 cycles per step, and the five
// sums of the ZBC epilogue made 60 of them per wave: tools/labs/fixed_cost_probe.py).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(unsigned long long)b, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double wave_sum_to_lane63(double v)
{
    v += dpp_mov<0xB1, 0xf>(v);
    v += dpp_mov<0x4E, 0xf>(v);
    v += dpp_mov<0x141, 0xf>(v);
    v += dpp_mov<0x140, 0xf>(v);
    v += dpp_mov<0x142, 0xa>(v);
    v += dpp_mov<0x143, 0xc>(v);
    return v;
}

// Sum over each DPP row of 16 lanes; EVERY lane of the row ends with its row's sum (the first four steps of
// wave_sum_to_lane63: 4 v_add_*_dpp for a float, 8 v_mov_b32_dpp + 4 v_add_f64 for a double).
__device__ __forceinline__ float row_sum16(float v)
{
    v += dpp_mov<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_mov<0x140, 0xf>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ double row_sum16(double v)
{
    v += dpp_mov<0xB1, 0xf>(v);
    v += dpp_mov<0x4E, 0xf>(v);
    v += dpp_mov<0x141, 0xf>(v);
    v += dpp_mov<0x140, 0xf>(v);
    return v;
}

// Per-wave accumulation of one value per maturity (the curve kernels: 100 reductions over the lanes inside the time
// loop, market_data.cuh:57-64).  The reference reduces the warp and lets lane 0 do a shared-memory atomic per
// maturity; the first version here reduced the wave to lane 63 (6 DPP steps) and had that lane read-modify-write a
// double in LDS (+ cvt + f64 add + the LDS round trip): ~50 of the ~85 issue cycles a maturity cost a wave.
// Now: 4 DPP steps give every lane its row-of-16 sum, and ONE register per wave collects 16 maturities: lane
// 16 r + i of `A` holds row r's running sum for maturity 16 g + i of the current group g (a single v_add under a
// lane mask).  Every 16th maturity the register is added, in double, to the wave's own LDS tile [groups][64] and
// cleared -- one LDS access per 16 maturities, no atomics, no cross-wave traffic, fixed order: bit-reproducible.
template <typename REAL>
struct MaturitySlots {
    REAL A;
    double* tile;      // this wave's [n_groups][64] doubles in LDS
    int lane, lane15;
    __device__ __forceinline__ void init(double* wave_tile)
    {
        A = 0;
        tile = wave_tile;
        lane = threadIdx.x & (kWave - 1);
        lane15 = lane & 15;
    }
    __device__ __forceinline__ void flush(int group)
    {
        tile[group * kWave + lane] += (double)A;
        A = 0;
    }
    // m (wave-uniform) in 1 .. n_mat-1; q: this lane's value for maturity m
    __device__ __forceinline__ void add(int ev, int m, int n_mat, const REAL& q)
    {
        if (ev == 0 || m >= n_mat) return;
        const REAL rs = row_sum16(q);
        if (lane15 == (m & 15)) A += rs;
        if ((m & 15) == 15) flush(m >> 4);
    }
    // after the last maturity of a path: the group in progress, unless add() has just flushed it
    __device__ __forceinline__ void finish(int n_mat)
    {
        if (((n_mat - 1) & 15) != 15) flush((n_mat - 1) >> 4);
    }
};
__host__ __device__ inline int maturity_groups(int n_mat) { return (n_mat + 15) >> 4; }

// The same job with a TRANSPOSED reduction: instead of reducing every maturity's 16 lane values to one number
// (4 DPP adds) and then picking one lane to keep it (compare + add + select), sixteen consecutive maturities are
// reduced TOGETHER, so that lane l of each row ends with the row sum of ONE of them.  Merging two vectors a (older
// maturity) and b (newer) at level L keeps a's half on the lanes whose class bit b_L is 0 and b's half on the others:
//     keep = b_L ? b : a;   send = b_L ? a : b;   merged = keep + dpp_L(send)          (2 v_cndmask + 1 v_add_dpp)
// and the merged vector represents both maturities in half as many lanes each.  The maturities arrive one at a time, so
// the pending vectors sit in a binary counter h0..h3: maturity k (= m mod 16) merges upwards through as many levels as
// k has trailing one bits -- 15 merges per 16 maturities, 2.9 VALU instructions per maturity instead of 7 -- and the
// sixteenth leaves one register that goes, in double, to the wave's LDS tile: one LDS access per 16 maturities.
// DPP partners: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (l -> l ^ 7), row_mirror (l -> l ^ 15); the class
// bits that make these four involutions a valid butterfly are b0 = l0^l2, b1 = l1^l2, b2 = l2^l3, b3 = l3 (l = lane & 15):
// each partner map flips exactly its own level's bit and none below it.  Lane l ends with maturity
// k = b0 + 2 b1 + 4 b2 + 8 b3 of the group; tree_lane_of() is the inverse.  Wave-uniform scalar branches only.
struct MaturityTree {
    float h0, h1, h2, h3;
    unsigned long long m0, m1, m2, m3;     // lane masks of the class bits (SGPR pairs)
    double* tile;      // this wave's [n_groups][64] doubles in LDS
    int lane;
    __device__ __forceinline__ void init(double* wave_tile)
    {
        tile = wave_tile;
        lane = threadIdx.x & (kWave - 1);
        const int l0 = lane & 1, l1 = (lane >> 1) & 1, l2 = (lane >> 2) & 1, l3 = (lane >> 3) & 1;
        m0 = __builtin_amdgcn_ballot_w64((l0 ^ l2) != 0);
        m1 = __builtin_amdgcn_ballot_w64((l1 ^ l2) != 0);
        m2 = __builtin_amdgcn_ballot_w64((l2 ^ l3) != 0);
        m3 = __builtin_amdgcn_ballot_w64(l3 != 0);
        h0 = h1 = h2 = h3 = 0.0f;   // maturity 0 is never simulated: level 0 starts with a zero vector pending
    }
    // m (wave-uniform) in 1 .. ; q: this lane's value for maturity m.
    // One asm statement: written in C++ the four pending vectors became phi nodes of the time loop and the compiler
    // paid for them with eight v_mov_b32 in EVERY Philox block (7 % of the loop); as tied asm operands they are updated
    // in place.  v_cndmask_b32 dst, a, b, mask = mask ? b : a;  v_add_f32_dpp dst, x, y = dpp(x) + y; a VALU result
    // needs two wait states before a DPP instruction may read it (s_nop 1).
    // `go` (wave-uniform): false = nothing to add in this call.  The test sits INSIDE the asm statement so that the
    // statement runs on every trip of the time loop: the pending vectors never pass through a phi node.
    // (The gate is passed as the integers it is made of, ev != 0 and m < n_mat: a bool would reach the asm through a
    // v_cndmask, i.e. in a vector register.)
    __device__ __forceinline__ void add(int ev, int m, int n_mat, float& v /* in: q; out: scratch */)
    {
        const int k = m & 15;
        float t;
        asm volatile(
            "s_cmp_eq_u32 %[ev], 0\n\t"
            "s_cbranch_scc1 9f\n\t"
            "s_cmp_ge_i32 %[m], %[nm]\n\t"
            "s_cbranch_scc1 9f\n\t"
            "s_bitcmp1_b32 %[k], 0\n\t"
            "s_cbranch_scc1 1f\n\t"
            "v_mov_b32 %[h0], %[v]\n\t"
            "s_branch 9f\n"
            "1:\n\t"
            "v_cndmask_b32_e64 %[t], %[v], %[h0], %[m0]\n\t"
            "v_cndmask_b32_e64 %[v], %[h0], %[v], %[m0]\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[v], %[t], %[v] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_bitcmp1_b32 %[k], 1\n\t"
            "s_cbranch_scc1 2f\n\t"
            "v_mov_b32 %[h1], %[v]\n\t"
            "s_branch 9f\n"
            "2:\n\t"
            "v_cndmask_b32_e64 %[t], %[v], %[h1], %[m1]\n\t"
            "v_cndmask_b32_e64 %[v], %[h1], %[v], %[m1]\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[v], %[t], %[v] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_bitcmp1_b32 %[k], 2\n\t"
            "s_cbranch_scc1 3f\n\t"
            "v_mov_b32 %[h2], %[v]\n\t"
            "s_branch 9f\n"
            "3:\n\t"
            "v_cndmask_b32_e64 %[t], %[v], %[h2], %[m2]\n\t"
            "v_cndmask_b32_e64 %[v], %[h2], %[v], %[m2]\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[v], %[t], %[v] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_bitcmp1_b32 %[k], 3\n\t"
            "s_cbranch_scc1 4f\n\t"
            "v_mov_b32 %[h3], %[v]\n\t"
            "s_branch 9f\n"
            "4:\n\t"
            "v_cndmask_b32_e64 %[t], %[v], %[h3], %[m3]\n\t"
            "v_cndmask_b32_e64 %[v], %[h3], %[v], %[m3]\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[v], %[t], %[v] row_mirror row_mask:0xf bank_mask:0xf\n"
            "9:\n\t"
            : [v] "+v"(v), [t] "=&v"(t), [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [h3] "+v"(h3)
            : [k] "s"(k), [ev] "s"(ev), [m] "s"(m), [nm] "s"(n_mat), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3)
            : "scc");
        if (ev != 0 && m < n_mat && k == 15) tile[(m >> 4) * kWave + lane] += (double)v;   // sixteen merged: lane l holds maturity class(l)
    }
    // after the last maturity of a path: complete the group in progress with zero vectors, re-arm for the next path
    __device__ __forceinline__ void finish(int n_mat)
    {
        for (int m = n_mat; (m & 15) != 0; m++) {
            float zero = 0.0f;
            add(1, m, m + 1, zero);
        }
        h0 = 0.0f;
    }
};
// lane (within a row of 16) that holds maturity k of a group once the tree has flushed it
__host__ __device__ inline int tree_lane_of(int k)
{
    const int l3 = (k >> 3) & 1, l2 = ((k >> 2) & 1) ^ l3, l1 = ((k >> 1) & 1) ^ l2, l0 = (k & 1) ^ l2;
    return l0 | (l1 << 1) | (l2 << 2) | (l3 << 3);
}

#ifndef HWK_MATURITY_TREE
#define HWK_MATURITY_TREE 1       // 0: MaturitySlots (A/B)
#endif
// the blocked float kernel uses the tree, the generic (double / odd stride) kernel the slots

// workgroup epilogue of the curve kernels: partials[block][k] = sum over waves and rows of the tiles' slot of maturity k
template <bool TREE>
__device__ __forceinline__ void maturity_tiles_to_partials(const double* s_tiles, int nwaves, int n_mat,
                                                           double* __restrict__ partials)
{
    const int groups = maturity_groups(n_mat);
    for (int k = threadIdx.x; k < n_mat; k += blockDim.x) {
        double tsum = 0.0;
        for (int w = 0; w < nwaves; w++) {
            const double* t = s_tiles + ((size_t)w * groups + (k >> 4)) * kWave + (TREE ? tree_lane_of(k & 15) : (k & 15));
            tsum += (t[0] + t[16]) + (t[32] + t[48]);
        }
        partials[(size_t)blockIdx.x * n_mat + k] = tsum;
    }
}

// Finish of a sum over the whole grid.  Every workgroup reduces its threads' N values (lane tree,
// waves in index order) and stores one row of partials.  What happens next is chosen by the host
// through the `ticket` argument:
//   ticket == nullptr : nothing more here; the host launches finalize_columns_kernel (the kernel
//                       boundary publishes the rows).  Default: simplest and, measured, fastest.
//   ticket != nullptr : single launch.  Rows are written with 8-byte agent-scope atomic stores
//                       (write-through, no release fence needed), the storing wave drains them
//                       (s_waitcnt vmcnt(0)), lane 0 takes a ticket on one of 8 sharded counters
//                       (blockIdx & 7, each on its own 128-B line); the last arriver of a shard takes
//                       a ticket on the top counter, and the workgroup that completes the top counter
//                       acquires (agent scope), reads all rows with agent-scope atomic loads, sums
//                       them in a FIXED order and re-arms the counters.  [2048 arrivals on ONE counter
//                       serialise at ~12 ns each = 25 us at the end of a 285 us kernel; sharding
//                       brings that to ~3 us.]
// Either way the order of the floating-point additions does not depend on timing: bit-reproducible.
constexpr int kTicketStride = 32;     // unsigned words between counters (128 B)

template <int N>
__device__ __forceinline__ void grid_sum_finish(double (&v)[N], double* smem /*[kMaxWaves*N]*/,
                                                double* __restrict__ partials, unsigned* __restrict__ ticket,
                                                double* __restrict__ out)
{
    __shared__ unsigned s_last;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int nwaves = (blockDim.x + kWave - 1) >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double s = wave_sum_to_lane63(v[k]);
        if (lane == kWave - 1) smem[wave * N + k] = s;
    }
    __syncthreads();
    if (wave != 0 && ticket == nullptr) return;
    if (wave == 0) {
        if (lane < N) {
            double t = 0.0;
            for (int w = 0; w < nwaves; w++) t += smem[w * N + lane];
            // row-major [grid][N]: the N doubles of a workgroup are contiguous (40 B for the five moments = two 32-byte
            // sectors; the column-major layout of round 1 dirtied one sector per value: WRITE_SIZE 320 KB for 82 KB)
            double* dst = partials + (size_t)blockIdx.x * N + lane;
            if (ticket == nullptr) *dst = t;
            else __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(t),
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ticket == nullptr) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave issued all the row's stores
        if (lane == 0) {
            const unsigned r = blockIdx.x & 7u;
            const unsigned n_sub = (gridDim.x - r + 7u) >> 3;      // workgroups with blockIdx & 7 == r
            const unsigned n_groups = gridDim.x < 8u ? gridDim.x : 8u;
            unsigned last = 0;
            const unsigned t1 = __hip_atomic_fetch_add(ticket + (1 + r) * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t1 == n_sub - 1) {
                const unsigned t2 = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (t2 == n_groups - 1) ? 1u : 0u;
            }
            s_last = last;
        }
    }
    __syncthreads();
    if (!s_last) return;                                           // workgroup-uniform
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    double acc[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const unsigned long long bits = __hip_atomic_load(reinterpret_cast<unsigned long long*>(partials + (size_t)b * N + k),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[k] += __longlong_as_double((long long)bits);
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) smem[wave * N + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double t = 0.0;
        for (int w = 0; w < nwaves; w++) t += smem[w * N + threadIdx.x];
        out[threadIdx.x] = t;
    }
    if (threadIdx.x < 9) __hip_atomic_store(ticket + threadIdx.x * kTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Second (tiny) launch: out[g*ncol + k] = scale[k] * sum_{b < nrows} partials[(g*nrows + b)*row_stride + k*col_stride],
// fixed order; grid = ncol * n_groups workgroups (n_groups > 1: several runs finished by one launch).
// All path kernels store row-major ([grid][N] for the moments, [grid][n_mat] for the curve: row_stride N, col_stride 1):
// a workgroup's row is written as one contiguous piece; the strided reads here are served by the L2.
// grid = ncol workgroups of 256 threads.  fix0: if >= 0, out[0] is overwritten with that value
// (P_sum[0] = 2 N of market_data.cuh:76-78, done without the reference's write race).
// stamps (may be nullptr): after its store every workgroup publishes `seq` in stamps[blockIdx.x] with system-scope release
// order -- when `out` is pinned host memory the host can see a result the moment it lands instead of waiting for the
// runtime to notice the end of the stream (hw_engine.hip, timed_sync).
__global__ void finalize_columns_kernel(const double* __restrict__ partials, int nrows, int row_stride, int col_stride,
                                        const double* __restrict__ scale, double fix0,
                                        double* __restrict__ out, int ncol,
                                        unsigned long long* __restrict__ stamps = nullptr, unsigned long long seq = 0)
{
    __shared__ double smem[kMaxWaves];
    const int k = blockIdx.x % ncol, group = blockIdx.x / ncol;
    const size_t row0 = (size_t)group * nrows;
    // four rows per trip with independent loads and accumulators (the 1024 rows of a 2^20-pair launch: ONE trip), fixed order
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int T = blockDim.x;
    for (int b = threadIdx.x; b < nrows; b += 4 * T) {
        const double* q = partials + (row0 + b) * row_stride + (size_t)k * col_stride;
        const size_t step = (size_t)T * row_stride;
        a0 += q[0];
        if (b + T < nrows) a1 += q[step];
        if (b + 2 * T < nrows) a2 += q[2 * step];
        if (b + 3 * T < nrows) a3 += q[3 * step];
    }
    double t = (a0 + a1) + (a2 + a3);
    // 256 threads = 4 waves: a DPP tree per wave (no LDS, no barrier), then ONE barrier and a fixed-order sum of the four
    // wave totals (round 4; rounds 1-3 ran an LDS tree of eight barriers: ~1 us of a 4 us kernel that every synchronous call waits for)
    t = wave_sum_to_lane63(t);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) smem[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int w = 0; w < (int)(blockDim.x + kWave - 1) / kWave; w++) r += smem[w];      // (the engine launches 256 threads: four waves)
        if (scale) r *= scale[k];
        if (k == 0 && fix0 >= 0.0) r = fix0;
        out[(size_t)group * ncol + k] = r;
        if (stamps != nullptr) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");             // system scope: the result before its stamp, as the host sees them
            __hip_atomic_store(stamps + blockIdx.x, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The shared inner loop: y_{i+1} = e y_i + G_i,  S_n = sum_{i=1..n} y_i, run on G / kBmScale: the y and S it
// returns are the true ones divided by kBmScale (see box_muller_unscaled).  r_i = m_i +- s y_i and
// int r = cI +- dt s (S - y/2) are reconstructed in the epilogues from host-computed m_n, cI_n (hw_model.hpp,
// mean_path).
//
// Per-step form (double; float when the (y, W) form below would lose precision): one fma + one add per step.
// float (y, W) form: a whole Philox block (4 steps) at a time, on the raw Box-Muller parts.  With g0 = s0 sin0,
// g1 = s0 cos0, g2 = s1 sin1, g3 = s1 cos1:
//     y4 = e^4 y0 + e^2 s0 (e sin0 + cos0) + s1 (e sin1 + cos1)
//     W4 = W0 + s0 (sin0 + cos0) + s1 (sin1 + cos1),            W_n = sum of the normals
// and, summing y_{i+1} - e y_i = g_i over the path,  S_n = (W_n - e y_n) / (1 - e)   (y_0 = 0).
// That is 10 single-rate ops per block instead of 12 (4 products, 4 fma, 4 adds) and two dependent fmas per
// block on y instead of four.  1 - e is exact in float (Sterbenz), and the rounding of W - e y, divided by
// 1 - e ~ 1e-2..2.5e-3, stays at ~2e-6 of S: the same size as the rounding of 500..2000 sequential adds.
// (Guard and measurements for weaker mean reversion: see the comment in ou_walk.)
// Experiments that were measured and NOT adopted are listed in DESIGN.md section 5; the A/B switches that remain
// (HWK_F64_BLOCK_FORM, HWK_ILP2_INTERLEAVE, HWK_MATURITY_TREE) select between two forms that are both tested.
// ------------------------------------------------------------------------------------------
#ifndef HWK_F64_BLOCK_FORM
#define HWK_F64_BLOCK_FORM 1     // 0: the double stepper runs one fma + one add per step on converted normals (A/B)
#endif
__device__ __forceinline__ void box_muller_parts(uint32_t x, uint32_t y, float& s, float& sn, float& cs)
{
    constexpr float kInv = 2.3283064e-10f;
    const float u = __builtin_fmaf((float)x, kInv, kInv);
    const float w = __builtin_fmaf((float)y, kInv, kInv);
    s = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u));
    sn = __builtin_amdgcn_sinf(w);
    cs = __builtin_amdgcn_cosf(w);
}

template <typename REAL, int ILP>
__device__ __forceinline__ void ou_walk(const PhiloxKey& key, const PhiloxPath (&pp)[ILP], int n_steps,
                                        REAL e, REAL (&y)[ILP], REAL (&S)[ILP])
{
    const int nblk = n_steps >> 2;
    const int rem = n_steps & 3;
    uint32_t a2s[ILP];
    bool uhi = true;                                  // wave-uniform: see philox_block_prio<UHI>
#pragma unroll
    for (int q = 0; q < ILP; q++) uhi = philox_uniform_hi(pp[q], a2s[q]) && uhi;
    HWK_SETPRIO(1);
    // The (y, W) form divides by 1 - e, which amplifies the rounding of W - e y by ~ 1 / ((1 - e) sqrt(n_steps)).
    // Measured against the double oracle (tools/labs/weak_reversion_probe.py): no excess over the per-step form at
    // (1 - e) sqrt(n) = 0.11 (a = 0.5), 1e-5 of excess at 0.022 (a = 0.1).  It is used from 0.05 up -- every
    // reference configuration: 0.22 at 1000 steps, 0.11 at 4000 -- weak mean reversion or very fine grids fall back
    // to the per-step form.
    bool wform = false;
    if constexpr (sizeof(REAL) == sizeof(float)) wform = (1.0f - (float)e) * sqrtf((float)n_steps) >= 0.05f;
    if (wform) {
      if constexpr (sizeof(REAL) == sizeof(float)) {
        const float e2 = e * e;
        float W[ILP];
#pragma unroll
        for (int q = 0; q < ILP; q++) W[q] = 0.0f;
        auto blocks_range = [&](auto tag, auto base_tag, int j0, int j1) {
            constexpr bool UHI = decltype(tag)::value;
            constexpr int BASE = decltype(base_tag)::value;
            for (int j = j0; j < j1; j++) {
#if HWK_ILP2_INTERLEAVE
                if constexpr (ILP == 2) {
                    uint32_t w[2][4];
                    philox_block_prio_x2<UHI>(key, pp, a2s, (uint32_t)j, w);
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        float s0, sn0, cs0, s1, sn1, cs1;
                        box_muller_parts(w[q][0], w[q][1], s0, sn0, cs0);
                        box_muller_parts(w[q][2], w[q][3], s1, sn1, cs1);
                        const float a0 = __builtin_fmaf(e, sn0, cs0), a1 = __builtin_fmaf(e, sn1, cs1);
                        const float b0 = sn0 + cs0, b1 = sn1 + cs1;
                        const float t = __builtin_fmaf(e2, y[q], s0 * a0);
                        y[q] = __builtin_fmaf(e2, t, s1 * a1);
                        W[q] = __builtin_fmaf(s1, b1, __builtin_fmaf(s0, b0, W[q]));
                    }
                    continue;
                }
#endif
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    uint32_t w0, w1, w2, w3;
                    float s0, sn0, cs0, s1, sn1, cs1;
                    philox_block_prio<UHI, true, BASE>(key, pp[q], a2s[q], (uint32_t)j, w0, w1, w2, w3);
                    box_muller_parts(w0, w1, s0, sn0, cs0);
                    box_muller_parts(w2, w3, s1, sn1, cs1);
                    const float a0 = __builtin_fmaf(e, sn0, cs0), a1 = __builtin_fmaf(e, sn1, cs1);
                    const float b0 = sn0 + cs0, b1 = sn1 + cs1;
                    const float t = __builtin_fmaf(e2, y[q], s0 * a0);
                    y[q] = __builtin_fmaf(e2, t, s1 * a1);
                    W[q] = __builtin_fmaf(s1, b1, __builtin_fmaf(s0, b0, W[q]));
                }
            }
        };
        auto blocks = [&](auto tag) {
            blocks_range(tag, std::integral_constant<int, 0>{}, 0, nblk);
        };
        if (uhi) blocks(std::true_type{});
        else blocks(std::false_type{});
        if (rem) {                                    // wave-uniform tail: 1..3 single steps
#pragma unroll
            for (int q = 0; q < ILP; q++) {
                float g[4];
                normal4_prio(key, pp[q], (uint32_t)nblk, g);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (c < rem) {
                        y[q] = __builtin_fmaf(y[q], e, g[c]);
                        W[q] += g[c];
                    }
                }
            }
        }
        const float inv = 1.0f / (1.0f - e);
#pragma unroll
        for (int q = 0; q < ILP; q++) S[q] = (W[q] - e * y[q]) * inv;
      }
    } else if (sizeof(REAL) == sizeof(double) && HWK_F64_BLOCK_FORM) {
        // double (y, W) form.  The same algebra as the float form above, with the recursion state in double: the
        // two block inputs  t0 = s0 (e sin0 + cos0),  t1 = s1 (e sin1 + cos1)  and the block's sum of normals
        // u = s0 (sin0 + cos0) + s1 (sin1 + cos1)  are formed in float from the raw Box-Muller parts (the normals are
        // float products anyway), converted once, and  y4 = e^2 (e^2 y0 + t0) + t1,  W += u  run in double:
        // 3 v_cvt_f64_f32 + 2 v_fma_f64 + 1 v_add_f64 per block instead of 4 + 4 + 4 (and no separate float
        // products for the normals).  The division by 1 - e at the end is harmless in double: no guard needed.
        const double e2 = (double)e * (double)e;
        const float ef = (float)e;
        double W[ILP];
#pragma unroll
        for (int q = 0; q < ILP; q++) W[q] = 0.0;
        auto blocks = [&](auto tag) {
            constexpr bool UHI = decltype(tag)::value;
            for (int j = 0; j < nblk; j++) {
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    uint32_t w0, w1, w2, w3;
                    float s0, sn0, cs0, s1, sn1, cs1;
                    philox_block_prio<UHI, true, 0>(key, pp[q], a2s[q], (uint32_t)j, w0, w1, w2, w3);
                    box_muller_parts(w0, w1, s0, sn0, cs0);
                    box_muller_parts(w2, w3, s1, sn1, cs1);
                    const float t0 = s0 * __builtin_fmaf(ef, sn0, cs0), t1 = s1 * __builtin_fmaf(ef, sn1, cs1);
                    const float u = __builtin_fmaf(s1, sn1 + cs1, s0 * (sn0 + cs0));
                    y[q] = (REAL)__builtin_fma(e2, __builtin_fma(e2, (double)y[q], (double)t0), (double)t1);
                    W[q] += (double)u;
                }
            }
        };
        if (uhi) blocks(std::true_type{});
        else blocks(std::false_type{});
        if (rem) {                                    // wave-uniform tail: 1..3 single steps
#pragma unroll
            for (int q = 0; q < ILP; q++) {
                float g[4];
                normal4_prio(key, pp[q], (uint32_t)nblk, g);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (c < rem) {
                        y[q] = hw_fma(y[q], e, (REAL)g[c]);
                        W[q] += (double)g[c];
                    }
                }
            }
        }
        const double inv = 1.0 / (1.0 - (double)e);
#pragma unroll
        for (int q = 0; q < ILP; q++) S[q] = (REAL)((W[q] - (double)e * (double)y[q]) * inv);
    } else {
        auto blocks = [&](auto tag) {
            constexpr bool UHI = decltype(tag)::value;
            for (int j = 0; j < nblk; j++) {
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    float g[4];
                    normal4_prio<UHI>(key, pp[q], a2s[q], (uint32_t)j, g);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        y[q] = hw_fma(y[q], e, (REAL)g[c]);
                        S[q] += y[q];
                    }
                }
            }
        };
        if (uhi) blocks(std::true_type{});
        else blocks(std::false_type{});
        if (rem) {                                    // wave-uniform tail
#pragma unroll
            for (int q = 0; q < ILP; q++) {
                float g[4];
                normal4_prio(key, pp[q], (uint32_t)nblk, g);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (c < rem) {
                        y[q] = hw_fma(y[q], e, (REAL)g[c]);
                        S[q] += y[q];
                    }
                }
            }
        }
    }
    HWK_SETPRIO(0);
}

// ------------------------------------------------------------------------------------------
// ZBC control-variate moments, superposed stepper.
// ------------------------------------------------------------------------------------------
template <typename REAL>
struct ZbcScen {
    REAL cz;   // dt * s * log2(e)          : exp(-+z),  z = dt s (S - y/2)
    REAL cw;   // B * s * log2(e)           : exp(-+w),  w = B s y
    REAL D0;   // exp(-cI_n)                : discount of the mean path
    REAL P0;   // A exp(-B m_n)             : bond price on the mean path
    REAL K;
};

template <typename REAL, int NSCEN>
struct ZbcArgs {
    int64_t n_pairs;
    uint64_t seed, path_offset;
    int32_t n_steps, iters;
    REAL e;
    ZbcScen<REAL> sc[NSCEN];
    unsigned long long* clk;   // optional clock probe (ClockProbe): 4 words per probing workgroup
    uint32_t clk_shift;        // probing workgroups: j << clk_shift, j < kClkProbes
    unsigned long long* stamps; // optional: every workgroup's entry/exit s_memrealtime (WorkgroupStamp)
    // several independent runs in one launch (the 20-run validation, 2_option:210-468): workgroup b belongs
    // to run b / blocks_per_run and uses seed + run * seed_stride.  One run: blocks_per_run = gridDim.x.
    uint32_t blocks_per_run;
    uint64_t seed_stride;
};

template <typename REAL, int NSCEN, int ILP>
__global__ void __launch_bounds__(1024)
zbc_superposed_kernel(const ZbcArgs<REAL, NSCEN> a, double* __restrict__ partials,
                      unsigned* __restrict__ ticket, double* __restrict__ out)
{
    __shared__ double smem[kMaxWaves * 5 * NSCEN];
    WorkgroupStamp wg_stamp;
    wg_stamp.begin(a.stamps);
    ClockProbe::begin(a.clk, a.clk_shift);          // diagnostic only; the stamps go to a buffer nothing else reads
    const uint32_t run = blockIdx.x / a.blocks_per_run;
    const uint32_t block_in_run = blockIdx.x - run * a.blocks_per_run;
    const PhiloxKey key = philox_key(a.seed + (uint64_t)run * a.seed_stride);
    PhiloxKey vkey = key;                           // round keys 3..10 in vector registers for the loop
    philox_key_to_vgprs(vkey);
    const int64_t T = (int64_t)a.blocks_per_run * blockDim.x;
    const int64_t t = (int64_t)block_in_run * blockDim.x + threadIdx.x;

    double acc[5 * NSCEN];
#pragma unroll
    for (int k = 0; k < 5 * NSCEN; k++) acc[k] = 0.0;

    for (int it = 0; it < a.iters; it++) {
        PhiloxPath pp[ILP];
        bool valid[ILP];
        REAL y[ILP], S[ILP];
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            int64_t p = t + ((int64_t)it * ILP + q) * T;
            valid[q] = p < a.n_pairs;
            if (!valid[q]) p = 0;
            pp[q] = philox_path(key, a.path_offset + (uint64_t)p);
            y[q] = 0;
            S[q] = 0;
        }
        ou_walk<REAL, ILP>(vkey, pp, a.n_steps, a.e, y, S);
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            const REAL h = S[q] - (REAL)0.5 * y[q];
#pragma unroll
            for (int k = 0; k < NSCEN; k++) {
                const ZbcScen<REAL>& sc = a.sc[k];
                const REAL z = sc.cz * h, w = sc.cw * y[q];
                const REAL d1 = sc.D0 * hw_exp2(-z), d2 = sc.D0 * hw_exp2(z);     // common.cuh:342-343
                const REAL P1 = sc.P0 * hw_exp2(-w), P2 = sc.P0 * hw_exp2(w);     // common.cuh:337-338
                const REAL c1 = d1 * P1, c2 = d2 * P2;                            // common.cuh:347-348
                const REAL x1 = d1 * hw_max0(P1 - sc.K), x2 = d2 * hw_max0(P2 - sc.K); // :352-353
                if (valid[q]) {
                    acc[5 * k + 0] += (double)(x1 + x2);                          // common.cuh:356-362
                    acc[5 * k + 1] += (double)(c1 + c2);
                    acc[5 * k + 2] += (double)(x1 * x1 + x2 * x2);
                    acc[5 * k + 3] += (double)(c1 * c1 + c2 * c2);
                    acc[5 * k + 4] += (double)(x1 * c1 + x2 * c2);
                }
            }
        }
    }
    ClockProbe::end(a.clk, a.clk_shift);
    grid_sum_finish<5 * NSCEN>(acc, smem, partials, ticket, out);
    // after the cross-wave finish (its barrier): the end of the workgroup's LAST wave.  (Stamped before it, thread 0 reports
    // when wave 0 -- the oldest of the workgroup, which the issue arbiter favours -- finished its paths: up to ~10 us early.)
    wg_stamp.end(a.stamps);
}

// ------------------------------------------------------------------------------------------
// ZBC control-variate moments, REFERENCE stepper: evolve_hull_white_step (common.cuh:237-244)
// verbatim on both legs, drift table read every step (scalar loads, or LDS when TABLE_LDS).
// float32 only (it exists to restate the reference's arithmetic on the GPU).
// ------------------------------------------------------------------------------------------
struct ZbcRefArgs {
    int64_t n_pairs;
    uint64_t seed, path_offset;
    int32_t n_steps, iters;
    float r0, e, dt, sig_st, A, B, K;
};

template <int TABLE_LDS>
__global__ void __launch_bounds__(1024)
zbc_reference_kernel(const ZbcRefArgs a, const float* __restrict__ drift, double* __restrict__ partials,
                     unsigned* __restrict__ ticket, double* __restrict__ out)
{
    __shared__ double smem[kMaxWaves * 5];
    extern __shared__ float s_drift[];
    const float* tab = drift;
    if (TABLE_LDS) {
        for (int i = threadIdx.x; i < a.n_steps; i += blockDim.x) s_drift[i] = drift[i];
        __syncthreads();
        tab = s_drift;
    }
    const PhiloxKey key = philox_key(a.seed);
    PhiloxKey vkey = key;                           // round keys 3..10 in vector registers for the loop
    philox_key_to_vgprs(vkey);
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr float kLog2e = 1.4426950408889634f;
    const float half_dt = 0.5f * a.dt;                  // exact (power-of-two scaling)
    // sig_st in a vector register: d + sig_st g contracts to v_fma(sig_st, g, d) with the wave-uniform drift d as its one
    // scalar operand; with sig_st scalar too, every step paid a v_mov to bring d into a vector register
    float sig_v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(sig_v) : "s"(a.sig_st));
    const float e_v = a.e;

    double acc[5] = {0, 0, 0, 0, 0};
    for (int it = 0; it < a.iters; it++) {
        int64_t p = t + (int64_t)it * T;
        const bool valid = p < a.n_pairs;
        if (!valid) p = 0;
        const PhiloxPath pp = philox_path(key, a.path_offset + (uint64_t)p);
        float r1 = a.r0, r2 = a.r0, I1 = 0.0f, I2 = 0.0f;           // common.cuh:318-319
        uint32_t a2s;
        const bool uhi = philox_uniform_hi(pp, a2s);                  // wave-uniform: scalar round-2 multiply
        HWK_SETPRIO(1);
        auto step = [&](float d, float gc) {
            const float sG = sig_v * gc;
            const float r1n = __builtin_fmaf(r1, e_v, d + sG);               // common.cuh:241
            I1 += (r1 + r1n) * half_dt;                                       // common.cuh:242; 0.5f*x*dt == x*(0.5f*dt) bit for bit
            r1 = r1n;
            const float r2n = __builtin_fmaf(r2, e_v, d - sG);
            I2 += (r2 + r2n) * half_dt;
            r2 = r2n;
        };
        auto walk = [&](auto tag) {
            constexpr bool UHI = decltype(tag)::value;
            const int nfull = a.n_steps & ~3;
            for (int i = 0; i < nfull; i += 4) {                     // common.cuh:325-332, four steps (one Philox block) per trip:
                float g[4];                                          // the four drifts come with ONE wave-uniform load
                normal4_prio<UHI>(vkey, pp, a2s, (uint32_t)(i >> 2), g);
                const float d0 = tab[i], d1 = tab[i + 1], d2 = tab[i + 2], d3 = tab[i + 3];
                step(d0, g[0]);
                step(d1, g[1]);
                step(d2, g[2]);
                step(d3, g[3]);
            }
            if (nfull < a.n_steps) {                                 // 1..3 remaining steps
                float g[4];
                normal4_prio<UHI>(vkey, pp, a2s, (uint32_t)(nfull >> 2), g);
#pragma unroll
                for (int c = 0; c < 3; c++)
                    if (nfull + c < a.n_steps) step(tab[nfull + c], g[c]);
            }
        };
        if (uhi) walk(std::true_type{});
        else walk(std::false_type{});
        HWK_SETPRIO(0);
        const float P1 = a.A * hw_exp2(-a.B * r1 * kLog2e);            // common.cuh:337-338
        const float P2 = a.A * hw_exp2(-a.B * r2 * kLog2e);
        const float d1 = hw_exp2(-I1 * kLog2e), d2 = hw_exp2(-I2 * kLog2e);  // common.cuh:342-343
        const float c1 = d1 * P1, c2 = d2 * P2;
        const float x1 = d1 * fmaxf(P1 - a.K, 0.0f), x2 = d2 * fmaxf(P2 - a.K, 0.0f);
        if (valid) {
            acc[0] += (double)(x1 + x2);
            acc[1] += (double)(c1 + c2);
            acc[2] += (double)(x1 * x1 + x2 * x2);
            acc[3] += (double)(c1 * c1 + c2 * c2);
            acc[4] += (double)(x1 * c1 + x2 * c2);
        }
    }
    grid_sum_finish<5>(acc, smem, partials, ticket, out);
}

// ------------------------------------------------------------------------------------------
// Zero-coupon curve: every `stride` steps add exp(-I1)+exp(-I2) = D0_m (2^{z'} + 2^{-z'}) to
// maturity m.  The lane-invariant factor D0_m is applied in double by finalize_columns_kernel.
// Per-wave running sums: MaturitySlots (one register per 16 maturities, flushed to the wave's LDS tile in double).
// ------------------------------------------------------------------------------------------
template <typename REAL>
struct ZcbArgs {
    int64_t n_pairs;
    uint64_t seed, path_offset;
    int32_t n_steps, n_mat, stride, iters;
    REAL e;
    REAL cz;   // dt * s * log2(e)
};

template <typename REAL>
__global__ void __launch_bounds__(1024)
zcb_curve_kernel(const ZcbArgs<REAL> a, double* __restrict__ partials)
{
    extern __shared__ double s_tiles[];                // [nwaves][groups][64], see MaturityTree
    const int wave = threadIdx.x >> 6;
    const int nwaves = (blockDim.x + kWave - 1) >> 6;
    const int groups = maturity_groups(a.n_mat);
    for (int i = threadIdx.x; i < nwaves * groups * kWave; i += blockDim.x) s_tiles[i] = 0.0;
    __syncthreads();
    MaturitySlots<REAL> slots;
    slots.init(s_tiles + (size_t)wave * groups * kWave);

    const PhiloxKey key = philox_key(a.seed);
    PhiloxKey vkey = key;                           // round keys 3..10 in vector registers for the loop
    philox_key_to_vgprs(vkey);
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;

    for (int it = 0; it < a.iters; it++) {
        int64_t p = t + (int64_t)it * T;
        const bool valid = p < a.n_pairs;
        if (!valid) p = 0;
        const PhiloxPath pp = philox_path(key, a.path_offset + (uint64_t)p);
        REAL y = 0, S = 0;
        uint32_t a2s;
        const bool uhi = philox_uniform_hi(pp, a2s);
        HWK_SETPRIO(1);
        auto walk = [&](auto tag) {
            constexpr bool UHI = decltype(tag)::value;
            int countdown = a.stride, m = 0;
            for (int i = 0; i < a.n_steps; i += 4) {
                float g[4];
                normal4_prio<UHI>(vkey, pp, a2s, (uint32_t)(i >> 2), g);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (i + c < a.n_steps) {                               // wave-uniform
                        y = hw_fma(y, a.e, (REAL)g[c]);
                        S += y;
                        if (--countdown == 0) {                            // market_data.cuh:57
                            countdown = a.stride;
                            m++;
                            if (m < a.n_mat) {                             // market_data.cuh:59
                                const REAL z = a.cz * (S - (REAL)0.5 * y);
                                REAL q = hw_exp2(z) + hw_exp2(-z);         // market_data.cuh:60
                                if (!valid) q = 0;
                                slots.add(1, m, a.n_mat, q);               // market_data.cuh:61-64
                            }
                        }
                    }
                }
            }
        };
        if (uhi) walk(std::true_type{});
        else walk(std::false_type{});
        HWK_SETPRIO(0);
        slots.finish(a.n_mat);
    }
    __syncthreads();
    maturity_tiles_to_partials<false>(s_tiles, nwaves, a.n_mat, partials);
}

// Blocked float form of the curve kernel (n_steps % 4 == 0, stride even and >= 4; everything else runs the
// generic kernel above).  Two changes against it:
//   * the steps advance half a Philox block (= one Box-Muller pair = 2 steps) at a time in the (y, W) form of
//     ou_walk: 5 single-rate ops per two steps instead of 6, and a maturity (every stride/2 half blocks) always
//     falls on a half-block boundary;  z = cz (S - y/2) = c1 W - c2 y with c1 = cz/(1-e), c2 = cz (e/(1-e) + 1/2);
//   * every thread carries P pairs through the time loop together, so the per-maturity reduction (MaturityTree:
//     ~3 instructions per maturity, one LDS access per 16 maturities) is paid once per P pairs: the P values are added
//     per lane first.
struct ZcbBlockedArgs {
    int64_t n_pairs;
    uint64_t seed, path_offset;
    int32_t n_steps, n_mat, stride, iters;
    float e, e2, c1, c2;
    unsigned long long* stamps;     // nullptr, or the timeline buffer (WorkgroupStamp)
    double e2_d, c1_d, c2_d;        // REAL = double: the same constants in double (e^2, c1, c2 from the double e)
    unsigned long long* clk;        // optional clock probe (ClockProbe), as ZbcArgs::clk / clk_shift
    uint32_t clk_shift;
};

// REAL = double (the double engine): the state (y, W) is carried in double -- the two half-block inputs s (e sin + cos)
// and s (sin + cos) are formed in float from the raw Box-Muller parts, as in the double stepper of ou_walk, converted
// once and accumulated with one v_fma_f64 / v_add_f64 each; z = c1 W - c2 y is formed in double and rounded to float
// for the two v_exp_f32 (their own error, <= 1 ulp of float and unbiased, is far below the 2e-6 floor the float normals
// set for the double engine); maturity sums as in the float kernel (rows of 16 in float, then double).
template <int P, typename REAL = float>
__global__ void __launch_bounds__(1024)
zcb_curve_blocked_kernel(const ZcbBlockedArgs a, double* __restrict__ partials)
{
    constexpr bool kDouble = std::is_same<REAL, double>::value;
    extern __shared__ double s_tiles[];                // [nwaves][groups][64], see MaturityTree
    WorkgroupStamp stamp;
    stamp.begin(a.stamps);
    ClockProbe::begin(a.clk, a.clk_shift);             // diagnostic: the clock this launch ran at
    const int wave = threadIdx.x >> 6;
    const int nwaves = (blockDim.x + kWave - 1) >> 6;
    const int groups = maturity_groups(a.n_mat);
    for (int i = threadIdx.x; i < nwaves * groups * kWave; i += blockDim.x) s_tiles[i] = 0.0;
    __syncthreads();
#if HWK_MATURITY_TREE
    MaturityTree slots;
#else
    MaturitySlots<float> slots;
#endif
    slots.init(s_tiles + (size_t)wave * groups * kWave);

    const PhiloxKey key = philox_key(a.seed);
    PhiloxKey vkey = key;
    philox_key_to_vgprs(vkey);
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nblk = a.n_steps >> 2, half_stride = a.stride >> 1;
    // e^2 multiplies y twice per block next to a VGPR partner (v_pk_fma_f32 {e2, s1} x {y, b1}): held in a vector
    // register it needs no v_mov per block to build that pair
    float e2v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(e2v) : "s"(a.e2));
    const float c1v = a.c1, c2v = a.c2;
    const float e1v = a.e;

    for (int it = 0; it < a.iters; it++) {
        PhiloxPath pp[P];
        REAL y[P], W[P];
        float live[P];
#pragma unroll
        for (int q = 0; q < P; q++) {
            int64_t p = t + ((int64_t)it * P + q) * T;
            live[q] = p < a.n_pairs ? 1.0f : 0.0f;
            if (p >= a.n_pairs) p = 0;
            pp[q] = philox_path(key, a.path_offset + (uint64_t)p);
            y[q] = 0;
            W[q] = 0;
        }
        uint32_t a2s[P];
        bool uhi = true;
#pragma unroll
        for (int q = 0; q < P; q++) uhi = philox_uniform_hi(pp[q], a2s[q]) && uhi;
        // Dead slots (pair index >= n_pairs) exist only in the last, ragged pass of the last workgroups; their contributions are
        // multiplied by live = 0.  Experiments (profiles/r02_curve_ab.txt): skipping that multiply for waves without dead
        // slots through a second copy of the loop measured no faster (0.3926 vs 0.3909 ms), and a
        // per-maturity scalar branch around it 2 % slower.
        HWK_SETPRIO(1);
        auto blocks = [&](auto tag) {
            constexpr bool UHI = decltype(tag)::value;
            int to_event = half_stride, m = 0;         // half blocks until the next maturity
            for (int j = 0; j < nblk; j++) {
                const int ev = to_event <= 2 ? to_event : 0;               // 1: after the first half, 2: after the second
                float accq;                                                // assigned by q == 0 of an event block
                auto contribute = [&](int q, REAL Wq) {
                    float z;
                    if constexpr (kDouble) z = (float)__builtin_fma(a.c1_d, Wq, -a.c2_d * y[q]);
                    else z = __builtin_fmaf(c1v, Wq, -c2v * y[q]);
                    const float c = hw_exp2(z) + hw_exp2(-z);              // market_data.cuh:60
                    accq = q == 0 ? live[q] * c : __builtin_fmaf(live[q], c, accq);
                };
#pragma unroll
                for (int q = 0; q < P; q++) {
                    uint32_t w0, w1, w2, w3;
                    philox_block_prio<UHI>(vkey, pp[q], a2s[q], (uint32_t)j, w0, w1, w2, w3);
                    float s0, sn0, cs0, s1, sn1, cs1;
                    box_muller_parts(w0, w1, s0, sn0, cs0);
                    box_muller_parts(w2, w3, s1, sn1, cs1);
                    // The event code is compared afresh (scalar, free) for every pair: left to itself the compiler computes
                    // `ev == 1` once, and -- every v_mad_u64_u32 in between clobbers an SGPR pair with its carry-out -- parks the
                    // flag in a VECTOR register (v_cndmask + v_cmp per pair: 1 % of the P = 4 loop, seen in the issue-bound twin's trace).
                    int evq = ev;
                    asm volatile("" : "+s"(evq));
                    // ev is wave-uniform: these are scalar branches.  The empty volatile asm keeps the compiler from
                    // if-converting them (it did: both pairs of v_exp_f32 ran in every event block and a v_cndmask
                    // picked one -- two transcendentals = 16 issue cycles wasted per maturity).
                    if constexpr (kDouble) {
                        const float t0 = s0 * __builtin_fmaf(e1v, sn0, cs0), t1 = s1 * __builtin_fmaf(e1v, sn1, cs1);
                        const float u0 = s0 * (sn0 + cs0), u1 = s1 * (sn1 + cs1);
                        // W advances by the block's sum of normals ONCE, in place, ahead of the event branches.  (Updated inside them --
                        // W += u0; contribute; W += u1 in one arm, W += u0 + u1 in the other -- the compiler hoisted the common sum into
                        // a temporary, because one arm still needed the old W, and paid a v_mov_b64 per pair and block for the copy
                        // back: issue-bound twin's trace.)  A maturity between the halves reads W - u1; the difference to W_old + u0
                        // is one rounding of u0 + u1 in float, 1e-7 of a normal, below the float rounding of z that follows.
                        W[q] += (double)(u0 + u1);
                        if (evq == 1) {                                    // a maturity between the two halves
                            asm volatile("" ::: "memory");
                            y[q] = __builtin_fma(a.e2_d, y[q], (double)t0);
                            contribute(q, W[q] - (double)u1);
                            y[q] = __builtin_fma(a.e2_d, y[q], (double)t1);
                        } else {                                           // whole block, as the double stepper of ou_walk:
                            y[q] = __builtin_fma(a.e2_d, __builtin_fma(a.e2_d, y[q], (double)t0), (double)t1);   // 3 cvt + 2 fma + 1 add
                            if (evq == 2) {
                                asm volatile("" ::: "memory");
                                contribute(q, W[q]);
                            }
                        }
                    } else {
                        y[q] = __builtin_fmaf(e2v, y[q], s0 * __builtin_fmaf(e1v, sn0, cs0));
                        W[q] = __builtin_fmaf(s0, sn0 + cs0, W[q]);
                        if (evq == 1) {
                            asm volatile("" ::: "memory");
                            contribute(q, W[q]);
                        }
                        y[q] = __builtin_fmaf(e2v, y[q], s1 * __builtin_fmaf(e1v, sn1, cs1));
                        W[q] = __builtin_fmaf(s1, sn1 + cs1, W[q]);
                        if (evq == 2) {
                            asm volatile("" ::: "memory");
                            contribute(q, W[q]);
                        }
                    }
                }
                m += ev != 0;
                slots.add(ev, m, a.n_mat, accq);                           // market_data.cuh:59-64 (no-op when ev == 0)
                to_event = ev ? half_stride - (2 - ev) : to_event - 2;
            }
        };
        if (uhi) blocks(std::true_type{});
        else blocks(std::false_type{});
        HWK_SETPRIO(0);
        slots.finish(a.n_mat);
    }
    ClockProbe::end(a.clk, a.clk_shift);
    __syncthreads();
    maturity_tiles_to_partials<HWK_MATURITY_TREE != 0>(s_tiles, nwaves, a.n_mat, partials);
    stamp.end(a.stamps);
}

// ------------------------------------------------------------------------------------------
// Pathwise Vega: d_sigma r follows the same OU recursion with its own drift and shock
// (s/sigma) G (3_sens:54-61), so  d_sigma r_i = D_i + (s/sigma) y_i  with the SAME y.
// ------------------------------------------------------------------------------------------
template <typename REAL>
struct VegaArgs {
    int64_t n_paths;
    uint64_t seed, path_offset;
    int32_t n_steps, iters, antithetic;
    REAL e;
    REAL cz, cw, D0, P0, K;   // as ZbcScen
    REAL B;                   // B(S1,S2)
    REAL gB;                  // sigma/(2a) (1 - e^{-2 a S1}) B      (3_sens:16-18)
    REAL Dn;                  // mean of d_sigma r at S1
    REAL cD;                  // mean of int d_sigma r
    REAL s_over_sigma;        // s / sigma
    REAL zs;                  // dt * s / sigma : stochastic part of int d_sigma r = zs (S - y/2)
    // several independent runs in one launch (the 20-run Vega bootstrap, 3_sens:527-654): workgroup b belongs to
    // run b / blocks_per_run and uses seed + run * seed_stride.  One run: blocks_per_run = gridDim.x.
    uint32_t blocks_per_run;
    uint64_t seed_stride;
};

template <typename REAL>
__device__ __forceinline__ REAL vega_leg(const VegaArgs<REAL>& a, REAL y, REAL h)
{
    const REAL z = a.cz * h, w = a.cw * y;
    const REAL Pv = a.P0 * hw_exp2(-w);                                    // 3_sens:65
    const REAL disc = a.D0 * hw_exp2(-z);                                  // 3_sens:68
    const REAL dsr = a.Dn + a.s_over_sigma * y;
    const REAL int_dsr = a.cD + a.zs * h;
    const REAL dP = -Pv * a.B * (a.gB + dsr);                              // 3_sens:15-19
    const REAL term1 = (Pv > a.K) ? dP * disc : (REAL)0;                   // 3_sens:72-74
    const REAL term2 = int_dsr * disc * hw_max0(Pv - a.K);                 // 3_sens:77-78
    return term1 - term2;                                                  // 3_sens:80
}

template <typename REAL, int ILP>
__global__ void __launch_bounds__(1024)
vega_pathwise_kernel(const VegaArgs<REAL> a, double* __restrict__ partials,
                     unsigned* __restrict__ ticket, double* __restrict__ out)
{
    __shared__ double smem[kMaxWaves];
    const uint32_t run = blockIdx.x / a.blocks_per_run;
    const uint32_t block_in_run = blockIdx.x - run * a.blocks_per_run;
    const PhiloxKey key = philox_key(a.seed + (uint64_t)run * a.seed_stride);
    PhiloxKey vkey = key;                           // round keys 3..10 in vector registers for the loop
    philox_key_to_vgprs(vkey);
    const int64_t T = (int64_t)a.blocks_per_run * blockDim.x;
    const int64_t t = (int64_t)block_in_run * blockDim.x + threadIdx.x;
    double acc[1] = {0.0};
    for (int it = 0; it < a.iters; it++) {
        PhiloxPath pp[ILP];
        bool valid[ILP];
        REAL y[ILP], S[ILP];
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            int64_t p = t + ((int64_t)it * ILP + q) * T;
            valid[q] = p < a.n_paths;
            if (!valid[q]) p = 0;
            pp[q] = philox_path(key, a.path_offset + (uint64_t)p);
            y[q] = 0;
            S[q] = 0;
        }
        ou_walk<REAL, ILP>(vkey, pp, a.n_steps, a.e, y, S);
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            const REAL h = S[q] - (REAL)0.5 * y[q];
            REAL v = vega_leg<REAL>(a, y[q], h);
            if (a.antithetic) v += vega_leg<REAL>(a, -y[q], -h);
            if (valid[q]) acc[0] += (double)v;
        }
    }
    grid_sum_finish<1>(acc, smem, partials, ticket, out);
}

// ------------------------------------------------------------------------------------------
// Sample trajectories for plotting (reference arithmetic, one thread per path, tiny).
// ------------------------------------------------------------------------------------------
__global__ void sample_paths_kernel(float* __restrict__ out, const float* __restrict__ drift, int n,
                                    int n_steps, uint64_t seed, uint64_t path_offset, float r0,
                                    float e, float sig_st)
{
    const int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= n) return;
    const PhiloxKey key = philox_key(seed);
    const PhiloxPath pp = philox_path(key, path_offset + (uint64_t)pid);
    float* row = out + (size_t)pid * (size_t)(n_steps + 1);
    float r = r0;
    row[0] = r;                                                            // market_data.cuh:148
    for (int i = 0; i < n_steps; i += 4) {
        float g[4];
        normal4(key, pp, (uint32_t)(i >> 2), g);
        for (int c = 0; c < 4 && i + c < n_steps; c++) {
            r = __builtin_fmaf(r, e, drift[i + c] + sig_st * g[c]);        // common.cuh:241
            row[i + c + 1] = r;                                            // market_data.cuh:157
        }
    }
}

// Device RNG probe: normals (hoisted Philox) of one subsequence, plus the raw words of every
// block computed twice -- raw[8j..8j+3] by the hoisted, prioritised form the kernels use, raw[8j+4..8j+7]
// by the plain ten-round form -- so a test can pin both against the oracle / rocRAND.
__global__ void debug_normals_kernel(float* __restrict__ out, uint32_t* __restrict__ raw, uint64_t seed,
                                     uint64_t subseq, int n_blocks)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_blocks) return;
    const PhiloxKey key = philox_key(seed);
    PhiloxKey vkey = key;                           // round keys 3..10 in vector registers for the loop
    philox_key_to_vgprs(vkey);
    const PhiloxPath pp = philox_path(key, subseq);
    // exactly what the hot loops consume: the prioritised block and the unscaled Box-Muller, rescaled here
    float g[4];
    normal4_prio(vkey, pp, (uint32_t)j, g);
    for (int c = 0; c < 4; c++) out[4 * j + c] = g[c] * (float)kBmScale;
    uint32_t o[4], h0, h1, h2, h3;
    philox_block_prio(vkey, pp, (uint32_t)j, h0, h1, h2, h3);
    HWK_SETPRIO(0);
    philox_block_plain(seed, subseq, (uint64_t)j, o);
    raw[8 * j + 0] = h0; raw[8 * j + 1] = h1; raw[8 * j + 2] = h2; raw[8 * j + 3] = h3;
    for (int c = 0; c < 4; c++) raw[8 * j + 4 + c] = o[c];
}

// Device probe of hw_exp2(double) (the double engine's exponential: v_exp_f32 + first-order correction)
__global__ void debug_exp2_kernel(const double* __restrict__ x, double* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hw_exp2(x[i]);
}

}  // namespace hwk
