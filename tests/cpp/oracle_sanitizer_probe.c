/* oracle_sanitizer_probe.c -- drives every entry point of the CPU oracle (oracle/hw_oracle.c) once, at small and
 * ragged sizes, so that a build with -fsanitize=address,undefined can vouch for the checker itself
 * ("host oracle under -fsanitize=address,undefined").  CPU only; tests/test_sanitizers.py
 * compiles this file together with hw_oracle.c and runs it -- never on the GPU box's card.
 * Prints one line of sums so the test can also compare the instrumented build with the normal one. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../../oracle/hw_oracle.h"

static double checksum(const double* v, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += v[i] * (1.0 + 0.001 * i);
    return s;
}

int main(void)
{
    ora_model m;
    if (ora_default_model(&m, "readme") != 0) return 2;
    if (ora_default_model(&m, "nope") == 0) return 3;
    if (ora_default_model(&m, "head") != 0) return 4;
    m.n_steps = 200;                 /* 20 maturities' worth of strides at n_mat = 21 */
    m.n_mat = 21;
    const int n = m.n_steps, nm = m.n_mat;

    float kf[6];
    double kd[6];
    ora_constants_f32(&m, m.sigma, kf);
    ora_constants_f64(&m, m.sigma, kd);

    float* df = malloc(sizeof(float) * n);
    float* sdf = malloc(sizeof(float) * n);
    float* shf = malloc(sizeof(float) * n);
    double* dd = malloc(sizeof(double) * n);
    double* sdd = malloc(sizeof(double) * n);
    double* shd = malloc(sizeof(double) * n);
    ora_tables_f32(&m, m.sigma, df, sdf);
    ora_tables_f64(&m, m.sigma, dd, sdd);
    ora_tables_f32(&m, m.sigma, df, NULL);            /* either table may be NULL */
    ora_tables_f64(&m, m.sigma, NULL, sdd);
    ora_shifted_drift_f32(&m, m.sigma + 1e-3, m.sigma, shf);
    ora_shifted_drift_f64(&m, m.sigma - 1e-3, m.sigma, shd);

    /* a smooth synthetic market curve on the n_mat grid */
    float* P = malloc(sizeof(float) * nm);
    float* f = malloc(sizeof(float) * nm);
    for (int j = 0; j < nm; j++) {
        const double T = m.T_final * j / (nm - 1);
        P[j] = (float)exp(-0.012 * T - 0.0007 * T * T);
        f[j] = (float)(0.012 + 0.0014 * T);
    }
    float abf[2];
    double abd[2];
    ora_bond_AB_f32(&m, m.sigma, P, f, 5.0, 10.0, abf);
    ora_bond_AB_f64(&m, m.sigma, P, f, 4.95, 9.99, abd);   /* off-grid: interpolation + clamp */

    uint32_t w[4];
    float g[4];
    ora_uint4(1234u, 0xFFFFFFFFFFull, 7u, w);
    ora_normal4(1234u, 3u, 0u, g);

    const double K = exp(-0.1);
    double z32[5], z64[5], zr[5], v32[1], v64[1];
    double* c32 = malloc(sizeof(double) * nm);
    double* c64 = malloc(sizeof(double) * nm);
    /* ragged counts (not multiples of the 512-pair chunk), 1 and 3 threads, a 64-bit path offset, zero pairs */
    if (ora_zbc_moments_f32(&m, m.sigma, df, P, f, 777, 1u, 0u, 5.0, 10.0, K, z32, 3)) return 5;
    if (ora_zbc_moments_f64(&m, m.sigma, dd, P, f, 513, 2u, 0x100000000ull - 100u, 5.0, 10.0, K, z64, 1)) return 6;
    if (ora_zbc_moments_f32(&m, m.sigma + 1e-3, shf, P, f, 0, 1u, 0u, 5.0, 10.0, K, zr, 2)) return 7;
    if (ora_zcb_curve_f32(&m, m.sigma, df, 600, 3u, 5u, c32, 3)) return 8;
    if (ora_zcb_curve_f64(&m, m.sigma, dd, 1, 4u, 0u, c64, 1)) return 9;
    if (ora_vega_pathwise_f32(&m, m.sigma, df, sdf, P, f, 1000, 5u, 0u, 5.0, 10.0, K, v32, 3)) return 10;
    if (ora_vega_pathwise_f64(&m, m.sigma, dd, sdd, P, f, 65, 6u, 9u, 5.0, 10.0, K, v64, 1)) return 11;
    {   /* the same body with the scale of its signed terms beside the sum */
        double t32[3], t64[3];
        if (ora_vega_pathwise_terms_f32(&m, m.sigma, df, sdf, P, f, 1000, 5u, 0u, 5.0, 10.0, K, t32, 3)) return 20;
        if (ora_vega_pathwise_terms_f64(&m, m.sigma, dd, sdd, P, f, 65, 6u, 9u, 5.0, 10.0, K, t64, 1)) return 21;
        if (t32[0] != v32[0] || t64[0] != v64[0] || !(t32[1] > 0.0) || !(t64[2] > 0.0)) return 22;
    }

    float* Pa = malloc(sizeof(float) * nm);
    float* fa = malloc(sizeof(float) * nm);
    ora_average_and_forward_f32(&m, c32, 1200, Pa, fa);
    float* paths = malloc(sizeof(float) * 3 * (n + 1));
    if (ora_sample_paths_f32(&m, m.sigma, df, 3, 9u, 0u, paths)) return 12;

    float cv32[9];
    double cv64[9];
    ora_cv_price_f32(z32, 2 * 777, P[nm - 1], cv32);
    ora_cv_price_f64(z64, 2 * 513, P[nm - 1], cv64);
    float* rec = malloc(sizeof(float) * nm);
    float* orig = malloc(sizeof(float) * nm);
    float* Ts = malloc(sizeof(float) * nm);
    ora_recover_theta_f32(&m, fa, rec, orig, Ts);
    float xs[20], st[8];
    for (int i = 0; i < 20; i++) xs[i] = 0.0354f + 1e-5f * (float)((i * 7) % 11);
    ora_run_statistics_f32(xs, 20, st);
    const int steps = ora_steps_to(&m, 5.0);

    printf("%.17g %.17g %.17g %.17g %.17g %.17g %.9g %.17g %.9g %d %d %u %.9g %.9g\n", checksum(z32, 5), checksum(z64, 5),
           checksum(c32, nm), checksum(c64, nm), v32[0], v64[0], cv32[6], cv64[6], st[4], steps, ora_max_threads(), w[0],
           g[0], paths[3 * (n + 1) - 1] + rec[nm - 1] + orig[0] + Ts[1] + kf[2] + (float)kd[3] + abf[0] + (float)abd[1] +
                     (float)zr[0]);

    free(df); free(sdf); free(shf); free(dd); free(sdd); free(shd); free(P); free(f); free(c32); free(c64);
    free(Pa); free(fa); free(paths); free(rec); free(orig); free(Ts);
    return 0;
}
