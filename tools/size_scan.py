#!/usr/bin/env python3
"""Batch-size scan of the path kernels on one GPU: kernel time against the number of pairs, 2^17 .. 2^26,
and a least-squares fit  t = t0 + c*n  over the sizes that fill the chip at least twice (the slope is the
asymptotic rate, t0 the fixed launch + fill/drain cost that a 2^20 batch cannot amortise).

    python tools/size_scan.py [--kernel zbc|vega|zcb] [--precision f32] [--grid-steps 1000] [--rounds 9]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hwmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="zbc", choices=["zbc", "zbc_ref", "vega", "zcb"])
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--grid-steps", type=int, default=1000)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--max-log2", type=int, default=26)
    a = ap.parse_args()
    p = hwmc.default_params("head", n_steps=a.grid_steps, precision=a.precision)
    e = hwmc.HullWhiteEngine(p, device=0)
    e.load_market_data(*hwmc.analytic_curve(p))
    sizes = []
    for k in range(17, a.max_log2 + 1):
        sizes.append(1 << k)
        if k < a.max_log2 and k < 23:
            sizes.append(3 << (k - 1))

    def run(n):
        e.profile_enable(1)
        if a.kernel == "zbc":
            e.simulate_ZBC_control_variate(n, 1234)
        elif a.kernel == "zbc_ref":
            e.simulate_ZBC_control_variate(n, 1234, stepper=hwmc.HW_STEPPER_REFERENCE)
        elif a.kernel == "vega":
            e.simulate_sensitivity(n, 1234)
        else:
            e.simulate_zcb(n, 1234)
        return float(e.profile_read(1)[0])

    t_end = time.time() + 0.5           # clock spin-up
    while time.time() < t_end:
        run(1 << 20)
    times = {n: [] for n in sizes}
    for _ in range(a.rounds):
        for n in sizes:
            run(n)                      # back-to-back pair: the second launch is the timed one
            times[n].append(run(n))
    per_thread = 2 if a.kernel in ("zbc", "zbc_ref", "zcb") else 1
    print(f"kernel {a.kernel}  precision {a.precision}  grid {a.grid_steps}  ({e.device_info()['name']})")
    print(f"{'pairs':>10} {'log2':>6} {'med ms':>9} {'min ms':>9} {'M paths/s (med)':>16}")
    med = {}
    for n in sizes:
        t = np.array(times[n])
        med[n] = float(np.median(t))
        print(f"{n:>10} {np.log2(n):>6.2f} {med[n]:>9.4f} {t.min():>9.4f} {per_thread * n / med[n] / 1e3:>16.1f}")
    fit = [n for n in sizes if n >= (1 << 20)]
    A = np.stack([np.ones(len(fit)), np.array(fit, float)], axis=1)
    t0, c = np.linalg.lstsq(A, np.array([med[n] for n in fit]), rcond=None)[0]
    print(f"fit over n >= 2^20:  t = {t0 * 1e3:.1f} us + n * {c * 1e6:.4f} ns   "
          f"-> asymptotic {per_thread / c / 1e3:.0f} M paths/s; fixed cost = {100 * t0 / med[1 << 20]:.1f} % of the 2^20 launch")


if __name__ == "__main__":
    main()
