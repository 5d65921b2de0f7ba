#!/usr/bin/env python3
"""ZBC kernel with one and with two pairs in flight per thread (hw_set_launch pairs_in_flight), kernel time at a few sizes.
    HWMC_LIB=... python tools/ilp_probe.py [grid steps = 1000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402

GRID = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
p = hwmc.default_params("head", n_steps=GRID)
e = hwmc.HullWhiteEngine(p, device=0)
e.load_market_data(*hwmc.analytic_curve(p))
for _ in range(300):
    e.simulate_ZBC_control_variate(1 << 20, 1)
ref = e.simulate_ZBC_control_variate(1 << 20, 7)
import time
for kernel in ("zbc", "vega"):
    fn = (lambda n: e.simulate_ZBC_control_variate(n, 1)) if kernel == "zbc" else (lambda n: e.simulate_sensitivity(n, 1))
    for ilp in (1, 2, 1, 2):
        e.set_launch(pairs_in_flight=ilp)
        assert np.allclose(e.simulate_ZBC_control_variate(1 << 20, 7), ref, rtol=1e-12)
        row = []
        for k in (20, 21, 22, 24, 26):
            n = 1 << k
            t0 = time.time()
            while time.time() - t0 < 0.12:
                fn(n)
            ms = []
            for _ in range(9 if k < 26 else 4):
                fn(n)
                e.profile_enable(1)
                fn(n)
                ms.append(float(e.profile_read(1)[0]))
            row.append("2^%d: %.4f" % (k, np.median(ms)))
        print("%s pairs in flight %d  " % (kernel, ilp) + "   ".join(row), flush=True)
