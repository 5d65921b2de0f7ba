#!/usr/bin/env python3
"""What does one maturity cost the curve kernel?  Same 2^k pairs x 1000 steps, the number of maturities varied
(n_mat - 1 = 1, 10, 50, 100, 250: stride 1000 .. 4 steps): a straight-line fit  t = t_step + n_events * c  separates
the pure stepping time (+ launch) from the per-maturity cost (2 exp2, the row reduction, the slot add).

    python tools/curve_event_cost.py [--log2 20] [--P 1]     (P: pairs per thread of the blocked kernel, via HWMC_CURVE_P)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--P", type=int, default=0)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--precision", default="f32")
    a = ap.parse_args()
    if a.P:
        os.environ["HWMC_CURVE_P"] = str(a.P)
    import hwmc
    n = 1 << a.log2
    rows = []
    for n_events in (1, 10, 50, 100, 250):
        p = hwmc.default_params("head", n_steps=1000, n_mat=n_events + 1, precision=a.precision)
        with hwmc.HullWhiteEngine(p, device=0) as e:
            for _ in range(40):
                e.simulate_zcb(n, 3)
            ms = []
            for _ in range(a.reps):
                e.simulate_zcb(n, 3)
                e.profile_enable(1)
                e.simulate_zcb(n, 3)
                ms.append(float(e.profile_read(1)[0]))
            rows.append((n_events, float(np.median(ms)), float(np.min(ms))))
    info = "2^%d pairs x 1000 steps, %s, P = %s" % (a.log2, a.precision, a.P or "engine default")
    print(info)
    print("%10s %10s %10s" % ("maturities", "med ms", "min ms"))
    for r in rows:
        print("%10d %10.4f %10.4f" % r)
    x = np.array([r[0] for r in rows], float)
    y = np.array([r[1] for r in rows])
    A = np.stack([np.ones(len(x)), x], axis=1)
    t0, c = np.linalg.lstsq(A, y, rcond=None)[0]
    waves = n / 64 / (a.P or 1)
    cyc = c * 1e-3 * 1024 * 2.4e9 / waves
    print("fit: t = %.1f us + maturities * %.3f us  -> %.0f SIMD issue cycles per maturity per wave (at 2.4 GHz, 1024 SIMDs);"
          " 100 maturities = %.1f %% of the stepping time" % (t0 * 1e3, c * 1e3, cyc, 100 * 100 * c / t0))


if __name__ == "__main__":
    main()
