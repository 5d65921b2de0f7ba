/*
 * hw_oracle.c -- CPU ORACLE (test infrastructure only; see hw_oracle.h for the rules,
 * the RNG stream definition and the parity-pinning statement).
 *
 * Build:  gcc -O3 -std=c11 -fopenmp -ffp-contract=off -shared -fPIC hw_oracle.c -o libhw_oracle.so -lm
 */
#include "hw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORA_CHUNK 512   /* pairs per partial sum: fixes the order of the floating-point additions (with or without OpenMP) */

/* ======================================================================================== *
 *  Philox4x32-10  (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11;
 *  constants and round function as published in Random123 and restated by rocRAND 4.2
 *  rocrand_philox4x32_10.h:62-65,325-345 -- third-party, not part of the reference).
 * ======================================================================================== */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void ora_uint4(uint64_t seed, uint64_t subseq, uint64_t block, uint32_t w[4])
{
    /* rocrand_init(seed, subsequence, offset=0) then `block` calls of rocrand4:
     * counter = {block lo, block hi, subseq lo, subseq hi} (rocrand_philox4x32_10.h:228-251). */
    const uint32_t ctr[4] = { (uint32_t)block, (uint32_t)(block >> 32),
                              (uint32_t)subseq, (uint32_t)(subseq >> 32) };
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    ora_philox4x32_10(ctr, key, w);
}

/* Box-Muller as rocrand_normal.h:53-68 (rocrand_device::detail::box_muller), with the two
 * uniform scalings written as explicit fused multiply-adds so the uniforms are bit-identical
 * to the GPU's (hipcc contracts the header's a + x*a into one v_fma_f32). */
#define TWO_POW32_INV     2.3283064e-10f   /* ROCRAND_2POW32_INV     */
#define TWO_POW32_INV_2PI 1.46291807e-09f  /* ROCRAND_2POW32_INV_2PI */

static inline void box_muller(uint32_t x, uint32_t y, float* n0, float* n1)
{
    const float u = fmaf((float)x, TWO_POW32_INV, TWO_POW32_INV);          /* (0, 1]   */
    const float v = fmaf((float)y, TWO_POW32_INV_2PI, TWO_POW32_INV_2PI);  /* (0, 2pi] */
    const float s = sqrtf(-2.0f * logf(u));
    *n0 = sinf(v) * s;
    *n1 = cosf(v) * s;
}

void ora_normal4(uint64_t seed, uint64_t subseq, uint64_t block, float g[4])
{
    uint32_t w[4];
    ora_uint4(seed, subseq, block, w);
    box_muller(w[0], w[1], &g[0], &g[1]);   /* rocrand_normal.h:259-269 normal_distribution4 */
    box_muller(w[2], w[3], &g[2], &g[3]);
}

/* ======================================================================================== */
int ora_default_model(ora_model* m, const char* profile)
{
    if (!m) return -1;
    m->a = 1.0; m->sigma = 0.1; m->r0 = 0.012; m->T_final = 10.0;   /* common.cuh:22,37-39 */
    m->n_steps = 1000; m->n_mat = 101;                              /* common.cuh:17,21    */
    m->th_break = 5.0; m->th_a1 = 0.012; m->th_b1 = 0.0014; m->th_b2 = 0.001; /* :228-230  */
    if (!profile || strcmp(profile, "head") == 0)        m->th_a2 = 0.019;  /* common.cuh:76,229 */
    else if (strcmp(profile, "readme") == 0)             m->th_a2 = 0.014;  /* 3_sens:387, README */
    else return -2;
    return 0;
}

int ora_steps_to(const ora_model* m, double S1)
{
    /* (int)(S1 / d_dt), common.cuh:322 -- evaluated as an integer ratio on the host so that it
     * does not depend on float-division rounding (500 for S1 = 5, T = 10, N = 1000). */
    const double x = S1 * (double)m->n_steps / m->T_final;
    return (int)floor(x + 1e-6);
}

int ora_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- float32 instantiation: the reference's arithmetic ---------------------------------- */
#define REAL float
#define R(x) x##f
#define SFX(name) name##_f32
#define R_EXP expf
#define R_SQRT sqrtf
#define R_COSH coshf
#define R_FMA fmaf
#define R_FMAX fmaxf
#include "hw_oracle_body.inc"
#undef REAL
#undef R
#undef SFX
#undef R_EXP
#undef R_SQRT
#undef R_COSH
#undef R_FMA
#undef R_FMAX

/* ---- float64 instantiation --------------------------------------------------------------- */
#define REAL double
#define R(x) x
#define SFX(name) name##_f64
#define R_EXP exp
#define R_SQRT sqrt
#define R_COSH cosh
#define R_FMA fma
#define R_FMAX fmax
#include "hw_oracle_body.inc"
#undef REAL
#undef R
#undef SFX
#undef R_EXP
#undef R_SQRT
#undef R_COSH
#undef R_FMA
#undef R_FMAX

/* ---- include/market_data.cuh:101-127 ----------------------------------------------------- */
void ora_average_and_forward_f32(const ora_model* m, const double* P_sum, int64_t n_total,
                                 float* P, float* f)
{
    const int n_mat = m->n_mat;
    const float inv_dT = 1 / ((float)m->T_final / (float)(n_mat - 1));   /* 1_bond:76 */
    for (int j = 0; j < n_mat; j++)
        P[j] = (float)P_sum[j] / (float)n_total;                         /* market_data.cuh:114 */
    for (int j = 0; j < n_mat; j++) {
        const int lo = (j == 0) ? 0 : j - 1;                             /* market_data.cuh:122 */
        const int hi = (j == n_mat - 1) ? n_mat - 1 : j + 1;             /* market_data.cuh:123 */
        const float scale = (j == 0 || j == n_mat - 1) ? 1.0f : 0.5f;    /* market_data.cuh:124 */
        f[j] = -scale * inv_dT * (logf(P[hi]) - logf(P[lo]));            /* market_data.cuh:125 */
    }
}

/* ---- include/market_data.cuh:136-160 ----------------------------------------------------- */
int ora_sample_paths_f32(const ora_model* m, double sigma, const float* drift,
                         int n, uint64_t seed, uint64_t path_offset, float* out)
{
    if (!m || !drift || !out || n < 0) return -1;
    float k[6];
    ora_constants_f32(m, sigma, k);
    const float dt = k[0], e = k[2], sig_st = k[3];
    const int n_steps = m->n_steps;
    for (int pid = 0; pid < n; pid++) {
        float r = (float)m->r0, integral = 0.0f, g[4];
        float* row = out + (size_t)pid * (size_t)(n_steps + 1);
        row[0] = r;                                                      /* market_data.cuh:148 */
        for (int i = 1; i <= n_steps; i++) {
            const int s = i - 1;
            if ((s & 3) == 0) ora_normal4(seed, path_offset + (uint64_t)pid, (uint64_t)(s >> 2), g);
            evolve_step_f32(&r, &integral, drift[s], sig_st * g[s & 3], e, dt);
            row[i] = r;                                                  /* market_data.cuh:157 */
        }
    }
    return 0;
}

/* ---- src/2_option_pricing.cu:14-35, include/common.cuh:228-230, :250-258 ------------------ */
void ora_recover_theta_f32(const ora_model* m, const float* f, float* theta_rec,
                           float* theta_orig, float* Ts)
{
    const int n = m->n_mat;
    const float a = (float)m->a, sigma = (float)m->sigma;
    const float spacing = (float)m->T_final / (float)(n - 1);
    for (int i = 0; i < n; i++) {
        const float T = (float)i * spacing;                              /* 2_option:25 */
        float d;
        if (i == 0)           d = (f[1] - f[0]) / spacing;               /* common.cuh:252 */
        else if (i == n - 1)  d = (f[i] - f[i - 1]) / spacing;           /* common.cuh:254 */
        else                  d = (f[i + 1] - f[i - 1]) / (2.0f * spacing); /* common.cuh:256 */
        const float convexity = (sigma * sigma / (2.0f * a)) * (1.0f - expf(-2.0f * a * T));
        theta_rec[i] = d + a * f[i] + convexity;                         /* 2_option:31 */
        theta_orig[i] = (T < (float)m->th_break) ? ((float)m->th_a1 + (float)m->th_b1 * T)
                                                 : ((float)m->th_a2 + (float)m->th_b2 * T);
        Ts[i] = T;
    }
}

/* ---- src/2_option_pricing.cu:305-324 (same arithmetic at 3_sens:570-589) ------------------ */
void ora_run_statistics_f32(const float* x, int n, float res[8])
{
    float mean = 0.0f;
    for (int i = 0; i < n; i++) mean += x[i];
    mean /= n;
    float variance = 0.0f;
    for (int i = 0; i < n; i++) { const float d = x[i] - mean; variance += d * d; }
    variance /= (n - 1);
    const float sd = sqrtf(variance);
    const float se = sd / sqrtf((float)n);
    const float t_critical = 2.093f;                                     /* 2_option:320 */
    const float margin = t_critical * se;
    res[0] = mean; res[1] = variance; res[2] = sd; res[3] = se; res[4] = margin;
    res[5] = mean - margin; res[6] = mean + margin; res[7] = 100.0f * sd / mean;
}
