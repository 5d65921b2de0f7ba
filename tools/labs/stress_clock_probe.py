#!/usr/bin/env python3
"""Which clock does a long (40-100 ms) launch really hold?  The ZBC kernel's in-kernel clock probe lives in workgroup 0, which
with the default geometry (one slot per pair) retires after the first ~0.3 ms of the launch; with a capped grid the threads
loop and workgroup 0 lives as long as the launch, like the blocked curve kernel's (grid capped by its wide rows).
    python tools/labs/stress_clock_probe.py [--log2 26] [--grid-steps 4000]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=26)
    ap.add_argument("--grid-steps", type=int, default=4000)
    a = ap.parse_args()
    n = 1 << a.log2
    for prec in ("f32", "f64"):
        p = hwmc.default_params("head", n_steps=a.grid_steps, precision=prec)
        with hwmc.HullWhiteEngine(p, device=0) as e:
            e.load_market_data(*hwmc.analytic_curve(p))
            info = e.device_info()
            simds = info["compute_units"] * 4

            def run(name, fn, steps):
                fn()
                e.profile_enable(3)
                clocks = []
                for _ in range(3):
                    fn()
                ms = [float(x) for x in e.profile_read(3)]
                mhz = e.profile_clock_mhz()
                m = float(np.median(ms))
                blocks = (n // 64) * (steps / 4.0) / simds
                print("%s %-34s %8.3f ms  %7.2f ns/block  clock %7.1f MHz  -> %7.2f cycles/block" %
                      (prec, name, m, m * 1e6 / blocks, mhz, m * 1e-3 * mhz * 1e6 / blocks), flush=True)

            half = a.grid_steps // 2
            run("zbc, default grid (probe: first slot)", lambda: e.simulate_ZBC_control_variate(n, 1234), half)
            for cap in (16, 4):
                e.set_launch(max_blocks_per_cu=cap)
                run("zbc, grid capped at %d wg/CU (whole launch)" % cap, lambda: e.simulate_ZBC_control_variate(n, 1234), half)
            e.set_launch(max_blocks_per_cu=0)
            run("curve (probe: whole launch)", lambda: e.simulate_zcb(n, 1234), a.grid_steps)


if __name__ == "__main__":
    main()
