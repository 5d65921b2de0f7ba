#!/usr/bin/env python3
"""Blocked float curve kernel against the generic per-step kernel at full size (both on the GPU):  max relative difference
of P_sum over the maturities, for several pair counts.   HWMC_LIB=... python tools/curve_selfcheck.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402

p = hwmc.default_params("readme")
os.environ["HWMC_CURVE_GENERIC"] = "1"
g = hwmc.HullWhiteEngine(p, device=0)
del os.environ["HWMC_CURVE_GENERIC"]
e = hwmc.HullWhiteEngine(p, device=0)
for n in (1000, 1 << 15, 1 << 17, 1 << 18, (1 << 18) + 512, 1 << 19, 1 << 20, (1 << 20) + 77, 1 << 21, 1 << 22):
    a, b = e.simulate_zcb(n, 99), g.simulate_zcb(n, 99)
    geo = e.launch_geometry("zcb_blocked", n)
    print("n = %8d  %s  max rel diff %.3e  (count exact: %s)" % (n, geo, float(np.abs(a[1:] / b[1:] - 1.0).max()), a[0] == b[0] == 2.0 * n))
