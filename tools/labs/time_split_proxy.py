#!/usr/bin/env python3
"""Upper bound on what splitting every path in TIME could gain an isolated 2^20-pair launch.

The proposal: hand the chip 2^21 half-paths (250 steps each) instead of 2^20 whole paths (500 steps), so that the last
round of waves is half as long and the launch drains through a thinning chip for half the time; the halves are joined by
y_500 = e^250 y_250 + tail, W additive, in the column pass (8 B per half-pair of extra traffic + the payoff there).

A path kernel that does exactly that cannot run faster than today's kernel given the SAME number of Philox blocks as 2^21
pairs x 250 steps: that launch has the half-length waves, the two rounds, the same per-wave prologue -- and, unlike the real
thing, no partial-state stores and no combine pass behind it.  So  T(2^21 pairs x 250 steps) / T(2^20 pairs x 500 steps)  is
a LOWER bound on the time ratio of the proposal; the acceptance bar was a gain of >= 3 %, i.e. a ratio <= 0.97.
    python tools/labs/time_split_proxy.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402

p = hwmc.default_params("head")
with hwmc.HullWhiteEngine(p, device=0) as e:
    e.load_market_data(*hwmc.analytic_curve(p))

    def ms_of(n, S1, reps=40):
        fn = lambda: e.simulate_ZBC_control_variate(n, 1234, S1=S1)
        for _ in range(30):
            fn()
        out = []
        for _ in range(reps):
            fn()
            e.profile_enable(1)
            fn()
            out.append(float(e.profile_read(1)[0]))
        return float(np.median(out)), float(np.min(out))

    print("isolated path-kernel time, hipEvent pair, median (min) of 40; same number of Philox blocks in every row")
    for ilp in (0, 1, 2):
        e.set_launch(pairs_in_flight=ilp if ilp else -1)
        whole = ms_of(1 << 20, 5.0)
        halves = ms_of(1 << 21, 2.5)
        quarters = ms_of(1 << 22, 1.25)
        tag = {0: "automatic (2 pairs per thread)", 1: "1 pair per thread", 2: "2 pairs per thread"}[ilp]
        print("%-32s 2^20 x 500 steps %.4f (%.4f) ms | 2^21 x 250 steps %.4f (%.4f) ms  ratio %.4f | 2^22 x 125 steps %.4f ms  ratio %.4f"
              % (tag, whole[0], whole[1], halves[0], halves[1], halves[0] / whole[0], quarters[0], quarters[0] / whole[0]), flush=True)
