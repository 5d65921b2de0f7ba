#!/usr/bin/env python3
"""Host-to-host time of synchronous hw_zbc_moments calls (2^20 pairs), the serial K = 20 loop and the column pass's share:
    python tools/labs/sync_call_now.py   (HWMC_LIB=/path/to/other/libhw_hip.so for an A/B)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402

p = hwmc.default_params("head")
with hwmc.HullWhiteEngine(p, device=0) as e:
    e.load_market_data(*hwmc.analytic_curve(p))
    req = e._request(1 << 20, 1234, 0, 5.0, 10.0, None, 0.0, hwmc.HW_DRIFT_BASE, hwmc.HW_STEPPER_SUPERPOSED)
    out5 = (C.c_double * 5)()
    call, ctx, rq = e.lib.hw_zbc_moments, e._ctx, C.byref(req)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        call(ctx, rq, out5, None)
    meds = []
    for rnd in range(7):
        ts = []
        for k in range(40):
            req.seed = 1234 + 12345 * k
            t0 = time.perf_counter()
            call(ctx, rq, out5, None)
            ts.append((time.perf_counter() - t0) * 1e3)
        meds.append(float(np.median(ts)))
    m = float(np.median(meds))
    print("synchronous call: median of 7 x 40 calls %.4f ms (rounds %s) = %.0f M paths/s = %.3fx the V100's 1064 M; sums[0] = %.6f"
          % (m, " ".join("%.4f" % x for x in meds), 2 * (1 << 20) / m / 1e3, 2 * (1 << 20) / m / 1e3 / 1064.0, out5[0]))
