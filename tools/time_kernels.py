#!/usr/bin/env python3
"""Steady-state kernel times of the three hot-path kernels (hipEvent pairs tight around the path
kernel, after a spin-up):  python tools/time_kernels.py [--log2 20] [--grid-steps 1000] [--precision f32]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hwmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--grid-steps", type=int, default=1000)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    n = 1 << a.log2
    p = hwmc.default_params("head", n_steps=a.grid_steps, precision=a.precision)
    e = hwmc.HullWhiteEngine(p, device=0)
    P, f = hwmc.analytic_curve(p)
    e.load_market_data(P, f)
    steps_s1 = a.grid_steps // 2
    jobs = {
        "zbc superposed": (lambda: e.simulate_ZBC_control_variate(n, 1), 2 * n, steps_s1),
        "zbc reference": (lambda: e.simulate_ZBC_control_variate(n, 1, stepper=hwmc.HW_STEPPER_REFERENCE), 2 * n, steps_s1),
        "zbc 3 scenarios": (lambda: e.simulate_ZBC_scenarios(n, 1, [0.099, 0.1, 0.101], [1, 0, 1]), 2 * n, steps_s1),
        "vega pathwise": (lambda: e.simulate_sensitivity(n, 1), n, steps_s1),
        "vega antithetic": (lambda: e.simulate_sensitivity(n, 1, antithetic=True), 2 * n, steps_s1),
        "zcb curve": (lambda: e.simulate_zcb(n, 1), 2 * n, a.grid_steps),
    }
    if a.precision == "f64":
        jobs.pop("zbc reference")
    t0 = time.time()
    while time.time() - t0 < 0.3:
        e.simulate_ZBC_control_variate(n, 1)
    print("%-18s %10s %10s %14s %20s" % ("kernel", "med ms", "min ms", "M paths/s", "G thread-steps/s"))
    for name, (fn, paths, steps) in jobs.items():
        t1 = time.time()
        while time.time() - t1 < 0.15:              # the chip's clock settles anew when the kernel changes: the first tens of
            fn()                                    # launches of another kernel measured up to 15 % slow (power management)
        ms = []
        for _ in range(a.reps):
            fn()                                    # back-to-back pair: the second launch is the timed one (a launch that
            e.profile_enable(1)                     # follows tens of microseconds of idle time -- reading the events back --
            fn()                                    # measured up to 15 % slower for the reference stepper)
            ms.append(float(e.profile_read(1)[0]))
        med = float(np.median(ms))
        units = paths if "vega pathwise" in name else paths // 2
        print("%-18s %10.4f %10.4f %14.1f %20.1f" % (name, med, min(ms), paths / med / 1000.0, units * steps / med / 1e6))


if __name__ == "__main__":
    main()
