#!/usr/bin/env python3
"""The four stress launches (2^26 pairs x 4000-step grid; ZBC stepper and curve kernel, float and double), six times each:
ms, launch-average in-kernel clock, cycles per pair-block -- run-to-run scatter of the figures bench.py's `stress` reports."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402
for prec in ("f32", "f64"):
    p = hwmc.default_params("head", n_steps=4000, precision=prec)
    with hwmc.HullWhiteEngine(p, device=0) as e:
        e.load_market_data(*hwmc.analytic_curve(p))
        n = 1 << 26
        for name, fn, steps in (("zbc", lambda: e.simulate_ZBC_control_variate(n, 1234), 2000), ("curve", lambda: e.simulate_zcb(n, 1234), 4000)):
            fn()
            rows = []
            for _ in range(6):
                e.profile_enable(1); fn(); ms = float(e.profile_read(1)[0]); mhz = e.profile_clock_mhz()
                rows.append((ms, mhz, ms * 1e-3 * mhz * 1e6 * 1024 / ((n // 64) * steps / 4.0)))
            print(prec, name, " ".join("%.2fms/%.0fMHz/%.1fcyc" % r for r in rows), flush=True)
