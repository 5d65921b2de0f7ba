#!/usr/bin/env python3
"""The numbers behind tests/test_gpu_readme_pin.py, printed: the reference's published 20-run validation (README.md:36-56)
repeated on the MI355X under theta profile "readme" with the README's own P(0,10) fed as P[100] / P0S2.

    python tools/readme_pin_report.py > profiles/rNN_readme_pin.txt
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hwmc  # noqa: E402

N = 1 << 20


def t_ci(x):
    x = np.asarray(x, float)
    return float(x.mean()), 2.093 * float(x.std(ddof=1)) / math.sqrt(len(x)), float(x.std(ddof=1))


def main():
    p = hwmc.default_params("readme")
    P, f = hwmc.analytic_curve(p)
    Pq = P.copy()
    Pq[100] = np.float32(0.876844)
    with hwmc.HullWhiteEngine(p, device=0) as e:
        e.load_market_data(Pq, f)
        sums, ms = e.simulate_ZBC_runs(N, 1234, n_runs=20, seed_stride=12345, return_ms=True)
        res = [hwmc.cv_price(s, 2 * N, 0.876844) for s in sums]
        price, ci, sd = t_ci([r["price"] for r in res])
        beta, bci, _ = t_ci([r["beta"] for r in res])
        rho = float(np.mean([r["rho"] for r in res]))
        print("device: %s" % e.device_info()["name"])
        print("20 runs x 2^20 pairs in one launch: %.3f ms" % ms)
        print("%-34s %-28s %s" % ("quantity", "MI355X (this build)", "reference README"))
        p10 = np.array([e.simulate_zcb(N, 1000 + 7919 * k)[100] / (2.0 * N) for k in range(16)])    # Q1, README.md:15-17
        print("%-34s %.6f +- %.6f (sd of one run, 16 seeds)   0.876844 (one run; exact 0.876777)" % ("Q1 P(0,10), one run of 2^20 pairs", p10.mean(), p10.std(ddof=1)))
        print("%-34s %.8f +- %.8f     0.03549203 +- 0.00000825" % ("ZBC price (95 % t-CI, 20 runs)", price, ci))
        print("%-34s %.4f %%                     0.0496 %%" % ("  CV of the 20 prices", 100 * sd / price))
        print("%-34s %.6f +- %.6f         0.166447 +- 0.000163" % ("beta*", beta, bci))
        print("%-34s %.4f                       0.673" % ("rho", rho))
        big = e.simulate_ZBC_runs(N, 777, n_runs=200, seed_stride=12345)
        r2 = [hwmc.cv_price(s, 2 * N, 0.876844) for s in big]
        adj = np.array([r["price"] for r in r2])
        raw = np.array([r["mean_X"] for r in r2])
        naive = np.array([r["mean_X"] - (r["mean_Y"] - 0.876844) for r in r2])
        print("%-34s %.3e (CV %.4f %%)      1.76e-05 (CV 0.0496 %%, 20 runs)" % ("sd of a run's price, 200 runs", adj.std(ddof=1), 100 * adj.std(ddof=1) / adj.mean()))
        print("%-34s %+.1f %% (200 runs)            +20.4 %% (20 runs)" % ("variance reduction, beta*", 100 * (1 - adj.var(ddof=1) / raw.var(ddof=1))))
        print("%-34s %+.1f %% (200 runs)            -42 %%" % ("variance reduction, beta = 1", 100 * (1 - naive.var(ddof=1) / raw.var(ddof=1))))
        runs = e.simulate_sensitivity_runs(N, 4242, n_runs=20, seed_stride=982451653) / N
        vega, vci, vsd = t_ci(runs)
        print("%-34s %.6f +- %.6f         0.230189 +- 0.000260" % ("pathwise Vega (20 runs)", vega, vci))
        print("%-34s %.3f %%                      0.24 %%" % ("  CV of the 20 Vegas", 100 * vsd / vega))
        fd = e.run_finite_difference(N, 4242 + 7)["vega_fd"]
        print("%-34s %.6f (%.2f %% from PW)     0.230316 (0.18 %%)" % ("finite-difference Vega (CRN)", fd, 100 * abs(fd - vega) / vega))
        rec = e.run_finite_difference_recalibrated(N, 99)["vega_fd_recalibrated"]
        print("%-34s %.6f (%.1f %% from PW)    0.523552 (127.74 %%)" % ("FD, curve re-simulated", rec, 100 * abs(rec - vega) / vega))
        print("exact Hull-White (SURVEY.md 6.2, readme profile): ZBC 0.03544557 (0.035492 with the README's P(0,10)), beta* 0.166309, rho 0.6738, Vega 0.230244")


if __name__ == "__main__":
    main()
