#!/usr/bin/env python3
"""Path-kernel time against the workgroup size (64 .. 1024 threads) at 2^20 pairs: hipEvent pair around the path kernel alone.
    python tools/block_sweep.py [--log2 20]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    a = ap.parse_args()
    n = 1 << a.log2
    p = hwmc.default_params("head")
    e = hwmc.HullWhiteEngine(p, device=0)
    e.load_market_data(*hwmc.analytic_curve(p))
    jobs = {"zbc": lambda: e.simulate_ZBC_control_variate(n, 1), "vega": lambda: e.simulate_sensitivity(n, 1), "zcb": lambda: e.simulate_zcb(n, 1)}
    for _ in range(300):
        jobs["zbc"]()
    print("%-6s %8s %10s %10s" % ("kernel", "block", "med ms", "min ms"))
    for name, fn in jobs.items():
        for bs in (64, 128, 256, 512, 1024):
            e.set_launch(block_size=bs)
            for _ in range(5):
                fn()
            ms = []
            for _ in range(15):
                fn()
                e.profile_enable(1)
                fn()
                ms.append(float(e.profile_read(1)[0]))
            print("%-6s %8d %10.4f %10.4f" % (name, bs, np.median(ms), min(ms)), flush=True)


if __name__ == "__main__":
    main()
