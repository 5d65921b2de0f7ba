#!/usr/bin/env python3
"""Measured parity envelope on the 24 random models of tests/test_gpu_parity.py::test_random_models_against_the_double_oracle:
float engine (both steppers) against the FLOAT oracle and the double oracle, the float oracle against the double one, and the
Vega error in units of sum|term1| + sum|term2|."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hwmc  # noqa: E402
import hw_oracle  # noqa: E402

hw_oracle.build()
rng = np.random.default_rng(20250607)
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(np.abs(np.asarray(b)), 1e-300)))
print("case  a     sigma  steps guard | sup-o32  ref-o32  sup-o64  o32-o64 | curve: sup-o32 o32-o64 | vega/scale: sup-o32 sup-o64 o32-o64 d64-o64")
worst = np.zeros(10)
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    n_mat = int(rng.choice([11, 21, 51, 101]))
    per = int(rng.choice([2, 3, 4, 5, 8, 10, 12]))
    n_steps = per * (n_mat - 1)
    T_final = float(rng.choice([5.0, 10.0, 20.0]))
    fields = dict(a=float(rng.uniform(0.3, 2.0)), sigma=float(rng.uniform(0.02, 0.2)), r0=float(rng.uniform(0.0, 0.05)),
                  T_final=T_final, th_break=float(rng.uniform(0.2, 0.8)) * T_final, th_a1=float(rng.uniform(0.0, 0.03)),
                  th_b1=float(rng.uniform(-0.001, 0.003)), th_a2=float(rng.uniform(0.0, 0.03)), th_b2=float(rng.uniform(-0.001, 0.003)))
    spacing = T_final / (n_mat - 1)
    i1 = int(rng.integers(1, n_mat - 1))
    S1, S2 = i1 * spacing, float(rng.uniform(i1 * spacing + 0.3 * spacing, T_final))
    engines, oracles = {}, {}
    for prec in ("f64", "f32"):
        p = hwmc.default_params("head", n_steps=n_steps, n_mat=n_mat, precision=prec)
        for k, v in fields.items():
            setattr(p, k, v)
        engines[prec] = hwmc.HullWhiteEngine(p, device=0)
        oracles[prec] = hw_oracle.Oracle("head", n_steps=n_steps, n_mat=n_mat, precision=prec)
        for k, v in fields.items():
            setattr(oracles[prec].m, k, v)
    P, f = hwmc.analytic_curve(engines["f64"].params)
    fwd = float(np.interp(S2, np.arange(n_mat) * spacing, P) / np.interp(S1, np.arange(n_mat) * spacing, P))
    K = fwd * float(rng.uniform(0.97, 1.03))
    n, seed = int(rng.integers(500, 6000)), int(rng.integers(1, 1 << 40))
    kw = dict(S1=S1, S2=S2, K=K)
    for e in engines.values():
        e.load_market_data(P, f)
    o32, o64 = oracles["f32"], oracles["f64"]
    w32, w64 = o32.zbc_moments(P, f, n, seed, **kw), o64.zbc_moments(P, f, n, seed, **kw)
    sup = engines["f32"].simulate_ZBC_control_variate(n, seed, **kw)
    ref = engines["f32"].simulate_ZBC_control_variate(n, seed, stepper=hwmc.HW_STEPPER_REFERENCE, **kw)
    c32, c64 = o32.zcb_curve(n, seed), o64.zcb_curve(n, seed)
    cs = engines["f32"].simulate_zcb(n, seed)
    v32, a1, a2 = o32.vega_pathwise_terms(P, f, n, seed, **kw)
    v64, b1, b2 = o64.vega_pathwise_terms(P, f, n, seed, **kw)
    scale = b1 + b2
    vs = engines["f32"].simulate_sensitivity(n, seed, **kw)
    vd = engines["f64"].simulate_sensitivity(n, seed, **kw)
    steps_s1 = o64.steps_to(S1)
    dt = T_final / n_steps
    guard = (1.0 - float(np.float32(np.exp(-fields["a"] * dt)))) * np.sqrt(steps_s1)
    row = [rel(sup, w32), rel(ref, w32), rel(sup, w64), rel(w32, w64), rel(cs, c32), rel(c32, c64),
           abs(vs - v32) / scale, abs(vs - v64) / scale, abs(v32 - v64) / scale, abs(vd - v64) / scale]
    worst = np.maximum(worst, row)
    print("%2d  %5.2f  %5.3f  %4d %5.3f | %.1e  %.1e  %.1e  %.1e | %.1e %.1e | %.1e %.1e %.1e %.1e   |v|/scale %.3f"
          % (case, fields["a"], fields["sigma"], steps_s1, guard, *row, abs(v64) / scale), flush=True)
    for e in engines.values():
        e.close()
print("worst:", " ".join("%.1e" % x for x in worst))
