"""Who loses precision when the mean reversion is weak?  f32/f64 engine, f32/f64 oracle and the reference stepper on
the same draws, relative to the f64 oracle, for a = 0.02, 0.1, 0.5 (test infrastructure: uses oracle/)."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hwmc, hw_oracle
for a in (0.02, 0.1, 0.5):
    res = {}
    for prec in ("f32", "f64"):
        p = hwmc.default_params("readme", precision=prec); p.a = a
        o = hw_oracle.Oracle("readme", precision=prec); o.m.a = a
        e = hwmc.HullWhiteEngine(p, device=0)
        P, f = hwmc.analytic_curve(p); e.load_market_data(P, f)
        res["e" + prec] = e.simulate_ZBC_control_variate(20000, 3)
        res["o" + prec] = o.zbc_moments(P, f, 20000, 3)
        if prec == "f32":
            res["eref"] = e.simulate_ZBC_control_variate(20000, 3, stepper=hwmc.HW_STEPPER_REFERENCE)
        e.close()
    ref = res["of64"]
    print("a =", a, {k: float(np.max(np.abs(v / ref - 1))) for k, v in res.items()})
