#!/usr/bin/env python3
"""Where the Monte Carlo error no longer hides a systematic one: the ZBC price of ONE call with 2^log2 pairs, float and double
engine on the SAME draws, against Jamshidian's closed form on the same market curve.

    python tools/bias_probe.py [--log2 33] > profiles/rNN_bias_probe.txt

float - double (same draws: no Monte Carlo noise) = what float32 stepping, payoff and accumulation cost;
reference stepper - double = the same for the reference's OWN per-step float arithmetic (evolve_hull_white_step verbatim,
common.cuh:237-244; HW_STEPPER_REFERENCE), so float - reference stepper is how far the superposed recursion sits from the
arithmetic a run of the reference performs on these draws;
double - closed form (+- SE) = what the grid costs (exact OU transition + trapezoid rule for the integral).
The north star's tolerance is 1e-5 of the price; the reference's own 20-run confidence interval is 2.3e-4 of it."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import hwmc
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=33)
    ap.add_argument("--seed", type=int, default=20240613)
    a = ap.parse_args()
    n = 1 << a.log2
    print("one call of 2^%d antithetic pairs (%.2e paths), seed %d; prices with the optimal-beta control variate" % (a.log2, 2.0 * n, a.seed))
    print("%-8s %5s %14s %14s %12s %10s %14s %12s %8s %14s %10s" % ("profile", "grid", "float", "double", "float-double", "rel", "closed form", "double-exact", "in SE",
                                                                  "ref.stepper-dbl", "rel"))
    for profile in ("readme", "head"):
        for grid in (1000, 4000):
            res = {}
            for prec in ("f32", "f64"):
                p = hwmc.default_params(profile, n_steps=grid, precision=prec)
                P, f = hwmc.analytic_curve(p)
                with hwmc.HullWhiteEngine(p, device=0) as e:
                    e.load_market_data(P, f)
                    t0 = time.perf_counter()
                    res[prec] = hwmc.cv_price(e.simulate_ZBC_control_variate(n, a.seed), 2 * n, float(P[-1]))
                    res[prec]["s"] = time.perf_counter() - t0
            pr = hwmc.default_params(profile, n_steps=grid, precision="f32", table_mode="ref_f32")   # the reference's float drift table too (common.cuh:60-84)
            with hwmc.HullWhiteEngine(pr, device=0) as e:
                e.load_market_data(P, f)
                ref = hwmc.cv_price(e.simulate_ZBC_control_variate(n, a.seed, stepper=hwmc.HW_STEPPER_REFERENCE), 2 * n, float(P[-1]))
            exact = hwmc.closed_form_zbc(p, P)
            se = res["f64"]["ci95_halfwidth"] / 1.96
            d32 = res["f32"]["price"] - res["f64"]["price"]
            dex = res["f64"]["price"] - exact
            dref = ref["price"] - res["f64"]["price"]
            print("%-8s %5d %14.10f %14.10f %12.2e %10.1e %14.10f %12.2e %8.2f %14.2e %10.1e   (SE %.2e; %.1f + %.1f s)"
                  % (profile, grid, res["f32"]["price"], res["f64"]["price"], d32, abs(d32) / exact, exact, dex, dex / se, dref, abs(dref) / exact, se,
                     res["f32"]["s"], res["f64"]["s"]), flush=True)


if __name__ == "__main__":
    main()
