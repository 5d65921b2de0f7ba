#!/usr/bin/env python3
"""Does the in-kernel clock probe of a launch fall with the launch's duration, and do the cycles per block it implies fall
with it?  Blocked curve kernel (P = 4 forced, grid capped by its wide rows: the clock probes sit in up to 64 workgroups spread over the grid, i.e. along the launch) and the
ZBC kernel with the grid capped at 16 workgroups per CU, 4000-step grid, 2^21..2^26 pairs, 6 launches back to back each."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["HWMC_CURVE_P"] = "4"
import hwmc  # noqa: E402

p = hwmc.default_params("head", n_steps=4000)
with hwmc.HullWhiteEngine(p, device=0) as e:
    e.load_market_data(*hwmc.analytic_curve(p))
    simds = e.device_info()["compute_units"] * 4
    for name, steps in (("curve", 4000), ("zbc", 2000)):
        for lg in (21, 22, 23, 24, 25, 26):
            n = 1 << lg
            if name == "zbc":
                e.set_launch(max_blocks_per_cu=16)
                fn = lambda: e.simulate_ZBC_control_variate(n, 1234)
            else:
                e.set_launch(max_blocks_per_cu=0)
                fn = lambda: e.simulate_zcb(n, 1234)
            fn()
            rows = []
            for _ in range(6):
                e.profile_enable(1)
                fn()
                ms = float(e.profile_read(1)[0])
                rows.append((ms, e.profile_clock_mhz()))
                samples = e.profile_clock_samples()
            ms = float(np.median([r[0] for r in rows]))
            mhz = float(np.median([r[1] for r in rows]))
            blocks = (n // 64) * (steps / 4.0) / simds
            print("%-5s 2^%d  %8.3f ms  %7.2f ns/block  clock %7.1f MHz (%s)  -> %7.2f cycles/block" %
                  (name, lg, ms, ms * 1e6 / blocks, mhz, " ".join("%.0f" % r[1] for r in rows), ms * 1e-3 * mhz * 1e6 / blocks), flush=True)
            print("        last launch, probing workgroups along the launch: " +
                  " ".join("%.0f@%.1f-%.1fms" % (m_, a_ / 1e3, b_ / 1e3) for m_, a_, b_ in zip(*samples)), flush=True)
