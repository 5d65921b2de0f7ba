// hw_model.hpp -- host-side model of the Hull-White path: step constants, drift tables and the
// deterministic ("mean path") part of the exact OU recursion, all evaluated once per launch on
// the host in double.  The reference computes the same tables in float32 host code and uploads
// them to __constant__ memory (include/common.cuh:60-110); the kernels here receive the few
// scalars they need as kernel arguments instead.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/hw_mc.h"

namespace hw {

struct StepConstants {
    double dt, spacing, e, sig_st, c1, c2;
};

// include/common.cuh:33-34, :87-96.  table_mode REF_F32 rounds every intermediate to float32
// exactly as the reference's host code does.
inline StepConstants step_constants(const hw_params& p, double sigma)
{
    StepConstants k{};
    if (p.table_mode == HW_TABLES_REF_F32) {
#pragma clang fp contract(off)
        const float a = (float)p.a, sg = (float)sigma;
        const float dt = (float)p.T_final / (float)p.n_steps;
        const float spacing = (float)p.T_final / (float)(p.n_mat - 1);
        const float e = expf(-a * dt);
        const float sig_st = sg * sqrtf((1.0f - expf(-2.0f * a * dt)) / (2.0f * a));
        const float c1 = (1.0f - e) / a;
        const float c2 = c1 / a;
        k = {dt, spacing, e, sig_st, c1, c2};
    } else {
        const double a = p.a;
        k.dt = p.T_final / p.n_steps;
        k.spacing = p.T_final / (p.n_mat - 1);
        k.e = std::exp(-a * k.dt);
        k.sig_st = sigma * std::sqrt(-std::expm1(-2.0 * a * k.dt) / (2.0 * a));
        k.c1 = -std::expm1(-a * k.dt) / a;
        k.c2 = k.c1 / a;
    }
    return k;
}

inline double theta(const hw_params& p, double t)
{
    return (t < p.th_break) ? (p.th_a1 + p.th_b1 * t) : (p.th_a2 + p.th_b2 * t);
}

// drift[i] = int_s^t e^{-a(t-u)} theta(u) du on step i (common.cuh:66-76);
// sigma_drift[i] = int_s^t e^{-a(t-u)} (sigma/a)(1 - e^{-2au}) du (common.cuh:79-80).
inline void drift_tables(const hw_params& p, double sigma, std::vector<double>* drift,
                         std::vector<double>* sigma_drift)
{
    const int n = p.n_steps;
    if (drift) drift->assign(n, 0.0);
    if (sigma_drift) sigma_drift->assign(n, 0.0);
    if (p.table_mode == HW_TABLES_REF_F32) {
#pragma clang fp contract(off)
        const StepConstants k = step_constants(p, sigma);
        const float a = (float)p.a, sg = (float)sigma, dt = (float)k.dt, e = (float)k.e;
        const float c1 = (float)k.c1, c2 = (float)k.c2, brk = (float)p.th_break;
        const float a1 = (float)p.th_a1, b1 = (float)p.th_b1, a2 = (float)p.th_a2, b2 = (float)p.th_b2;
        for (int i = 0; i < n; i++) {
            const float s = (float)i * dt;
            const float t = (float)(i + 1) * dt;
            const float first = ((s + dt) - e * s) / a - c2;
            if (drift) (*drift)[i] = (s < brk) ? (b1 * first + a1 * c1) : (b2 * first + a2 * c1);
            if (sigma_drift) {
                const float st = (2.0f * sg * expf(-a * t)) * (coshf(a * t) - coshf(a * s));
                (*sigma_drift)[i] = st / (a * a);
            }
        }
        return;
    }
    const double a = p.a, dt = p.T_final / n;
    const double e = std::exp(-a * dt);
    const double c1 = -std::expm1(-a * dt) / a, c2 = c1 / a;
    for (int i = 0; i < n; i++) {
        const double s = i * dt, t = (i + 1) * dt;
        // int_s^t e^{-a(t-u)} u du = (t - e s)/a - (1-e)/a^2
        const double first = (t - e * s) / a - c2;
        if (drift) (*drift)[i] = (s < p.th_break) ? (p.th_b1 * first + p.th_a1 * c1)
                                                  : (p.th_b2 * first + p.th_a2 * c1);
        if (sigma_drift) {
            // (sigma/a) [ (1-e)/a - e^{-at}(e^{at}-e^{as}) ... ] written without cancellation:
            // int_s^t e^{-a(t-u)}(1-e^{-2au}) du = (1-e)/a - e^{-a t}(e^{-a s}-e^{-a t})/a
            const double v = c1 - std::exp(-a * t) * (std::exp(-a * s) - std::exp(-a * t)) / a;
            (*sigma_drift)[i] = (sigma / a) * v;
        }
    }
}

// compute_shifted_drift_table, src/3_sensitivity_analysis.cu:374-398: base drift (fitted at
// sigma_old) + the change of the convexity term sigma^2/(2a)(1-e^{-2au}) integrated over the step.
inline void shifted_drift_table(const hw_params& p, double sigma_new, double sigma_old,
                                std::vector<double>* out)
{
    std::vector<double> base;
    drift_tables(p, sigma_old, &base, nullptr);
    const int n = p.n_steps;
    out->assign(n, 0.0);
    if (p.table_mode == HW_TABLES_REF_F32) {
#pragma clang fp contract(off)
        const float a = (float)p.a, dt = (float)p.T_final / (float)p.n_steps;
        const float sn = (float)sigma_new, so = (float)sigma_old;
        const float shift = (sn * sn - so * so) / (2.0f * a);
        for (int i = 0; i < n; i++) {
            const float s = (float)i * dt, t = (float)(i + 1) * dt;
            const float adj = (shift / a) * (1.0f + expf(-2.0f * a * t) - expf(-a * (t - s)) -
                                             expf(-a * (t + s)));
            (*out)[i] = (float)base[i] + adj;
        }
        return;
    }
    const double a = p.a, dt = p.T_final / n;
    const double shift = (sigma_new * sigma_new - sigma_old * sigma_old) / (2.0 * a);
    for (int i = 0; i < n; i++) {
        const double s = i * dt, t = (i + 1) * dt;
        // (1 - e^{-a dt}) - e^{-a(t+s)}(1 - e^{-a dt}) = (1 - e^{-a dt})(1 - e^{-a(t+s)})
        const double adj = (shift / a) * (-std::expm1(-a * dt)) * (-std::expm1(-a * (t + s)));
        (*out)[i] = base[i] + adj;
    }
}

// (int)(S1/dt) of common.cuh:322, as an integer ratio.
inline int steps_to(const hw_params& p, double S1)
{
    return (int)std::floor(S1 * (double)p.n_steps / p.T_final + 1e-6);
}

// Linear superposition of the exact OU recursion  r_{i+1} = e r_i + drift_i + s G_i :
//   r_i = m_i + s y_i,   m_{i+1} = e m_i + drift_i (m_0 = r0),   y_{i+1} = e y_i + G_i (y_0 = 0).
// The trapezoid integral of common.cuh:242 telescopes to
//   I_n = dt (sum_{i=1..n} r_i + (r_0 - r_n)/2) = cI_n + dt s (Sy_n - y_n/2),
//   cI_n = dt (M_n + (r0 - m_n)/2),  M_n = sum_{i=1..n} m_i,  Sy_n = sum_{i=1..n} y_i.
// mean_path fills m[i], cI[i] for i = 0..n.
inline void mean_path(double x0, double e, double dt, const std::vector<double>& drift, int n,
                      std::vector<double>* m, std::vector<double>* cI)
{
    m->assign(n + 1, 0.0);
    cI->assign(n + 1, 0.0);
    double mi = x0, M = 0.0;
    (*m)[0] = mi;
    for (int i = 0; i < n; i++) {
        mi = std::fma(mi, e, drift[i]);
        M += mi;
        (*m)[i + 1] = mi;
        (*cI)[i + 1] = dt * (M + 0.5 * (x0 - mi));
    }
}

// include/common.cuh:187-196
inline double interpolate(const std::vector<float>& data, double T, double spacing)
{
    const int n_mat = (int)data.size();
    int idx = (int)std::floor(T / spacing + 1e-9);
    if (idx >= n_mat - 1) return data[n_mat - 1];
    const double alpha = (T - idx * spacing) / spacing;
    return data[idx] * (1.0 - alpha) + data[idx + 1] * alpha;
}

// A(t,T), B(t,T): include/common.cuh:180-182, :201-214.
inline void bond_AB(const hw_params& p, double sigma, const std::vector<float>& P,
                    const std::vector<float>& f, double t, double T, double* A, double* B)
{
    const double a = p.a, spacing = p.T_final / (p.n_mat - 1);
    const double Bv = -std::expm1(-a * (T - t)) / a;
    const double P0T = interpolate(P, T, spacing), P0t = interpolate(P, t, spacing);
    const double f0t = interpolate(f, t, spacing);
    const double term3 = (sigma * sigma / (4.0 * a)) * (-std::expm1(-2.0 * a * t)) * Bv * Bv;
    *A = (P0T / P0t) * std::exp(Bv * f0t - term3);
    *B = Bv;
}

// Exact curve: P(0,T) = exp(-M(T) + V(T)/2), f(0,T) = r0 e^{-aT} + int_0^T theta e^{-a(T-u)} du
// - sigma^2 B(0,T)^2 / 2, for piecewise-linear theta.
inline void analytic_curve(const hw_params& p, double T, double* P, double* f)
{
    const double a = p.a, sg = p.sigma;
    const double B0T = -std::expm1(-a * T) / a;
    // piece integrals of theta(u) e^{-a(T-u)} and theta(u) B(u,T) over [u0,u1]
    auto piece = [&](double al, double be, double u0, double u1, double* ie, double* ib) {
        if (u1 <= u0) { *ie = 0; *ib = 0; return; }
        const double E1 = std::exp(-a * (T - u1)), E0 = std::exp(-a * (T - u0));
        const double i_exp = (E1 - E0) / a;                                         // int e^{-a(T-u)}
        const double i_uexp = (u1 / a - 1 / (a * a)) * E1 - (u0 / a - 1 / (a * a)) * E0; // int u e^{..}
        *ie = al * i_exp + be * i_uexp;
        const double i_lin = al * (u1 - u0) + be * (u1 * u1 - u0 * u0) / 2.0;
        *ib = (i_lin - *ie) / a;
    };
    const double tb = std::fmin(T, p.th_break);
    double ie1, ib1, ie2, ib2;
    piece(p.th_a1, p.th_b1, 0.0, tb, &ie1, &ib1);
    piece(p.th_a2, p.th_b2, p.th_break, T, &ie2, &ib2);
    const double M = p.r0 * B0T + ib1 + ib2;
    const double V = sg * sg / (a * a) * (T - 2.0 * B0T + (-std::expm1(-2.0 * a * T)) / (2.0 * a));
    *P = std::exp(-M + 0.5 * V);
    *f = p.r0 * std::exp(-a * T) + ie1 + ie2 - 0.5 * sg * sg * B0T * B0T;
}

}  // namespace hw
