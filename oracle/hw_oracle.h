/*
 * hw_oracle.h -- CPU ORACLE for the Hull-White Monte Carlo hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped engine (libhw_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * What it is: a plain-C restatement of the reference's algorithm
 * (giulialionetti/Monte-Carlo-simulation-of-Hull-White-model-and-sensitivities-computation),
 * function by function, each citing the reference file:line it follows
 * (paths relative to the reference root).  The reference's float32 arithmetic is
 * restated in float32 ("_f32" entry points, IEEE, no contraction); "_f64" entry points
 * run the same algorithm in double.
 *
 * Random numbers.  The reference draws N(0,1) with cuRAND XORWOW seeded from time(NULL)
 * (include/common.cuh:277-280, src/2_option_pricing.cu:128), so no reproducible
 * reference stream exists.  This build's stream (oracle and GPU alike) is
 * Philox4x32-10 laid out exactly as rocRAND 4.2's rocrand_state_philox4x32_10:
 *   key     = {seed lo32, seed hi32}
 *   counter = {block j, 0, subsequence lo32, subsequence hi32},  subsequence = global path id
 *   normals = rocrand_normal4 Box-Muller of the four output words, in order x,y,z,w;
 *   time step i (0-based) of a path consumes component (i & 3) of block (i >> 2).
 * Philox is pinned by the Random123 known-answer vectors (tests/test_oracle_philox.py) and by
 * rocRAND's own host-callable implementation (tests/golden/gen_rocrand_vectors.cpp).
 *
 * PARITY PINNING.  The reference holds no golden vectors, unit tests or fixed seeds (SURVEY.md
 * section 4), and its CUDA sources cannot be compiled here (no nvcc/cuRAND), so bit-level
 * parity with a reference run is UNPINNED at the RNG boundary.  The oracle is pinned instead by
 * (1) the deterministic constants/tables of the reference's host code (SURVEY.md section 2.3 KATs),
 * (2) exact Hull-White closed forms, and (3) the reference's published outputs
 * (README.md:15-56) as statistical targets under theta profile "readme".
 *
 * SUMMATION ORDER.  The Monte Carlo sums are formed per fixed chunk of 512 pairs (ORA_CHUNK) and the chunks are
 * added in index order, so every result is bit-identical whatever the number of OpenMP threads or the schedule
 * (tests/golden/gen_golden.py regenerates the committed fixture byte for byte).
 */
#ifndef HW_ORACLE_H
#define HW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Model description; defaults = include/common.cuh:16-39 and theta_func (common.cuh:228-230). */
typedef struct ora_model {
    double a;           /* H_A      = 1.0   */
    double sigma;       /* H_SIGMA  = 0.1   */
    double r0;          /* H_R0     = 0.012 */
    double T_final;     /* T_FINAL  = 10.0  */
    int32_t n_steps;    /* N_STEPS  = 1000  */
    int32_t n_mat;      /* N_MAT    = 101   */
    /* theta(t) = th_a1 + th_b1 t  (t < th_break) ;  th_a2 + th_b2 t  (t >= th_break) */
    double th_break, th_a1, th_b1, th_a2, th_b2;
} ora_model;

/* profile: "head" -> th_a2 = 0.019 (source as mounted), "readme" -> 0.014 (published numbers). */
int  ora_default_model(ora_model* m, const char* profile);

/* ---- RNG ------------------------------------------------------------------------------- */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* The four normals of Philox block `block` of subsequence `subseq` under `seed`. */
void ora_normal4(uint64_t seed, uint64_t subseq, uint64_t block, float g[4]);
/* The four raw 32-bit words of that block. */
void ora_uint4(uint64_t seed, uint64_t subseq, uint64_t block, uint32_t w[4]);

/* ---- constants and tables (include/common.cuh:60-110) ---------------------------------- */
/* step constants: out[0]=dt, [1]=mat_spacing, [2]=exp(-a dt), [3]=sig_st(sigma),
 * [4]=(1-e)/a, [5]=(1-e)/a^2 */
void ora_constants_f32(const ora_model* m, double sigma, float out[6]);
void ora_constants_f64(const ora_model* m, double sigma, double out[6]);
/* drift[n_steps], sigma_drift[n_steps] for volatility sigma. */
void ora_tables_f32(const ora_model* m, double sigma, float* drift, float* sigma_drift);
void ora_tables_f64(const ora_model* m, double sigma, double* drift, double* sigma_drift);
/* FD drift table shifted for sigma_new, src/3_sensitivity_analysis.cu:374-398.
 * (the reference hard-codes the README-era 0.014 on the s>=5 branch; here the model's th_a2
 *  is used on that branch -- unreachable for S1 <= th_break.) */
void ora_shifted_drift_f32(const ora_model* m, double sigma_new, double sigma_old, float* drift);
void ora_shifted_drift_f64(const ora_model* m, double sigma_new, double sigma_old, double* drift);
/* number of executed steps to reach S1: (int)(S1/dt) of common.cuh:322, computed robustly. */
int  ora_steps_to(const ora_model* m, double S1);

/* A(t,T), B(t,T) of common.cuh:180-214 on the n_mat-point market curves. out[0]=A, out[1]=B */
void ora_bond_AB_f32(const ora_model* m, double sigma, const float* P, const float* f,
                     double S1, double S2, float out[2]);
void ora_bond_AB_f64(const ora_model* m, double sigma, const float* P, const float* f,
                     double S1, double S2, double out[2]);

/* ---- hot-path steppers ----------------------------------------------------------------- */
/* All take `drift` (and `sigma_drift`) tables explicitly (the reference keeps them in
 * mutable __constant__ symbols, which Q3 overwrites between launches) and a `sigma`.
 * n_threads: OpenMP threads (1 = scalar loop).  Return 0 on success.                      */

/* simulate_ZBC_control_variate, include/common.cuh:286-409.
 * out[5] = {sum X, sum Y, sum X^2, sum Y^2, sum XY} over 2*n_pairs samples. */
int ora_zbc_moments_f32(const ora_model* m, double sigma, const float* drift,
                        const float* P, const float* f,
                        int64_t n_pairs, uint64_t seed, uint64_t path_offset,
                        double S1, double S2, double K, double out[5], int n_threads);
int ora_zbc_moments_f64(const ora_model* m, double sigma, const double* drift,
                        const float* P, const float* f,
                        int64_t n_pairs, uint64_t seed, uint64_t path_offset,
                        double S1, double S2, double K, double out[5], int n_threads);

/* simulate_zcb, include/market_data.cuh:25-79.  P_sum[n_mat]; P_sum[0] = 2*n_pairs. */
int ora_zcb_curve_f32(const ora_model* m, double sigma, const float* drift,
                      int64_t n_pairs, uint64_t seed, uint64_t path_offset,
                      double* P_sum, int n_threads);
int ora_zcb_curve_f64(const ora_model* m, double sigma, const double* drift,
                      int64_t n_pairs, uint64_t seed, uint64_t path_offset,
                      double* P_sum, int n_threads);

/* compute_average_and_forward, include/market_data.cuh:101-127 (n_total = 2*n_pairs). */
void ora_average_and_forward_f32(const ora_model* m, const double* P_sum, int64_t n_total,
                                 float* P, float* f);

/* simulate_sensitivity, src/3_sensitivity_analysis.cu:15-96.  out[0] = sum(term1 - term2). */
int ora_vega_pathwise_f32(const ora_model* m, double sigma, const float* drift,
                          const float* sigma_drift, const float* P, const float* f,
                          int64_t n_paths, uint64_t seed, uint64_t path_offset,
                          double S1, double S2, double K, double out[1], int n_threads);
int ora_vega_pathwise_f64(const ora_model* m, double sigma, const double* drift,
                          const double* sigma_drift, const float* P, const float* f,
                          int64_t n_paths, uint64_t seed, uint64_t path_offset,
                          double S1, double S2, double K, double out[1], int n_threads);

/* The same loop with the scale of the sum beside it: out[0] = sum(term1 - term2), out[1] = sum |term1|, out[2] = sum |term2|
 * (3_sens:72-80; the two absolute sums are not in the reference: they scale the tests' error bounds for a sum of signed terms). */
int ora_vega_pathwise_terms_f32(const ora_model* m, double sigma, const float* drift,
                                const float* sigma_drift, const float* P, const float* f,
                                int64_t n_paths, uint64_t seed, uint64_t path_offset,
                                double S1, double S2, double K, double out[3], int n_threads);
int ora_vega_pathwise_terms_f64(const ora_model* m, double sigma, const double* drift,
                                const double* sigma_drift, const float* P, const float* f,
                                int64_t n_paths, uint64_t seed, uint64_t path_offset,
                                double S1, double S2, double K, double out[3], int n_threads);

/* simulate_paths_show, include/market_data.cuh:136-160.  out[n][n_steps+1], row-major. */
int ora_sample_paths_f32(const ora_model* m, double sigma, const float* drift,
                         int n, uint64_t seed, uint64_t path_offset, float* out);

/* ---- host post-processing -------------------------------------------------------------- */
/* beta*, CV price: src/2_option_pricing.cu:154-175,277-281 in float32.
 * res = {mean_X, mean_Y, var_Y, cov, beta, adjustment, price, var_X, rho} */
void ora_cv_price_f32(const double sums[5], int64_t n_total, double P0S2, float res[9]);
void ora_cv_price_f64(const double sums[5], int64_t n_total, double P0S2, double res[9]);

/* recover_theta, src/2_option_pricing.cu:14-35 + theta_func common.cuh:228-230. */
void ora_recover_theta_f32(const ora_model* m, const float* f, float* theta_rec,
                           float* theta_orig, float* Ts);

/* 20-run statistics, src/2_option_pricing.cu:305-324: res = {mean, var(unbiased), sd, se,
 * margin (t=2.093), ci_lo, ci_hi, cv_percent} */
void ora_run_statistics_f32(const float* x, int n, float res[8]);

int ora_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
