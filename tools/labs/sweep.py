#!/usr/bin/env python3
"""Launch-geometry / stepper sweep of the ZBC kernel on the GPU it runs on (tuning harness).

Replaces the reference's block-size sweep (src/3_sensitivity_analysis.cu:277-371).  Every variant
is timed interleaved in ONE process (hipEvent pairs tight around the path kernel), median and min
over the rounds are reported.   python tools/sweep.py [--log2 20] [--rounds 7] [--grid-steps 1000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--grid-steps", type=int, default=1000)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--kernel", default="zbc", choices=["zbc", "vega", "zcb"])
    a = ap.parse_args()
    n = 1 << a.log2
    p = hwmc.default_params("head", n_steps=a.grid_steps, precision=a.precision)
    e = hwmc.HullWhiteEngine(p, device=0)
    P, f = hwmc.analytic_curve(p)
    e.load_market_data(P, f)
    print(e.device_info())
    for k in ("zbc", "zbc_ref", "zcb", "zcb_blocked", "vega"):
        print(k, e.kernel_info(k))
    variants = []
    for stepper in ("superposed", "reference"):
        if a.kernel != "zbc" and stepper == "reference":
            continue
        for block in (128, 256, 512, 1024):
            for cap in (0, 2 * max(1, 32 // (block // 64)), 1 << 20):
                for ilp in (1, 2):
                    for lds in ((0, 1) if stepper == "reference" else (0,)):
                        if stepper == "reference" and ilp == 2:
                            continue
                        variants.append((stepper, block, cap, ilp, lds))
    times = {v: [] for v in variants}

    def run(v):
        stepper, block, cap, ilp, lds = v
        e.set_launch(block, cap, lds, ilp)
        e.profile_enable(1)
        if a.kernel == "zbc":
            st = hwmc.HW_STEPPER_SUPERPOSED if stepper == "superposed" else hwmc.HW_STEPPER_REFERENCE
            e.simulate_ZBC_control_variate(n, 1234, stepper=st)
        elif a.kernel == "vega":
            e.simulate_sensitivity(n, 1234)
        else:
            e.simulate_zcb(n, 1234)
        return float(e.profile_read(1)[0])

    for v in variants:      # warm-up
        run(v)
    for _ in range(a.rounds):
        for v in variants:
            times[v].append(run(v))
    paths = (2 * n) if a.kernel != "vega" else n
    print("%-11s %5s %8s %3s %3s %9s %9s %12s" % ("stepper", "block", "cap/CU", "ilp", "lds", "med ms", "min ms", "Mpaths/s(med)"))
    for v in sorted(variants, key=lambda v: np.median(times[v])):
        t = np.array(times[v])
        print("%-11s %5d %8d %3d %3d %9.4f %9.4f %12.1f" % (v[0], v[1], v[2], v[3], v[4], np.median(t), t.min(),
                                                            paths / np.median(t) / 1000.0))


if __name__ == "__main__":
    main()
