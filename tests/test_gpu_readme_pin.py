"""The only pin the REFERENCE itself holds for this path, applied to the GPU engine at the reference's own sample size.

The reference ships no golden vectors and seeds everything from time(NULL) with XORWOW (2_option_pricing.cu:128,223), so
no bit-level comparison with a reference run exists ("parity unpinned" at the RNG boundary).  What
it does publish is the OUTPUT of its own 20-run statistical validation (README.md:15-17, 36-56), produced with the
README-era theta (intercept 0.014 after t = 5: theta profile "readme"):

    P(0,10) = 0.876844                         (Q1, README.md:16)
    ZBC     = 0.03549203 +- 0.00000825  95 % t-CI over 20 runs of 2^20 pairs, CV 0.0496 %   (README.md:41-42)
    beta*   = 0.166447 +- 0.000163, rho = 0.673, variance reduction +20.4 %                  (README.md:36-39)
    Vega    = 0.230189 +- 0.000260 (20 runs, CV 0.24 %); single runs PW 0.229895, FD 0.230316 (0.18 %)   (README.md:49-56)

This test repeats that validation on the MI355X -- 20 batched runs of 2^20 antithetic pairs, seeds base + run*12345
(2_option:229) and base + run*982451653 (3_sens:545) -- and asks for agreement within the COMBINED 95 % intervals plus
the part of the reference's input that the README does not publish.

What cannot be pinned.  The reference's Q2/Q3 consumed the curve its own Q1 had just simulated (data/P.bin, f.bin).  Of
that curve the README prints only P(0,10) = 0.876844 (fed here as P[100] and P0S2, exactly as 2_option:492 does) and
f(0,0) = 1.21 %.  The option uses three more numbers of it -- P[50], f[50] (in A(5,10), common.cuh:201-214) -- which are
not published; the exact curve stands in for them.  Their Monte Carlo error in the reference's run is bounded by what
IS published: |P_MC(0,10) - P_exact(0,10)| = 6.7e-5 for a bond price and |f_MC(0,0) - 0.012| <= 1.5e-4 for a forward rate
(1.21 % printed to two decimals).  The budget for a quantity Q is |dQ/dP50| * 6.7e-5 + |dQ/df50| * 1.5e-4 with the
derivatives measured here by common-random-number repricing.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1 << 20                      # N_PATHS of the reference (common.cuh:16)
README = {
    "P10": 0.876844, "zbc": 0.03549203, "zbc_ci": 0.00000825, "zbc_cv_pct": 0.0496,
    "beta": 0.166447, "beta_ci": 0.000163, "rho": 0.673, "var_reduction_pct": 20.4,
    "vega20": 0.230189, "vega20_ci": 0.000260, "vega_cv_pct": 0.24, "vega_pw": 0.229895, "vega_fd": 0.230316,
}
U_P, U_F = 6.7e-5, 1.5e-4        # published size of the reference's Q1 Monte Carlo error (see the module docstring)


@pytest.fixture(scope="module")
def setup(hwmc_mod):
    p = hwmc_mod.default_params("readme")
    P, f = hwmc_mod.analytic_curve(p)
    assert abs(float(P[100]) - 0.876777) < 2e-6                      # exact P(0,10) under the README-era theta
    Pq = P.copy()
    Pq[100] = np.float32(README["P10"])                               # the published Q1 estimate the README's Q2/Q3 consumed
    e = hwmc_mod.HullWhiteEngine(p, device=0)
    e.load_market_data(Pq, f)
    yield hwmc_mod, e, Pq, f
    e.close()


def _t_ci(x):
    x = np.asarray(x, float)
    return float(x.mean()), 2.093 * float(x.std(ddof=1)) / math.sqrt(len(x))          # 2_option:305-324


def _curve_budget(hwmc, e, Pq, f, quantity):
    """|dQ/dP50| U_P + |dQ/df50| U_F by central differences with common random numbers (one 2^20-pair run each)."""
    out = []
    for idx_arr, h in (("P", 2e-4), ("f", 2e-4)):
        vals = []
        for sgn in (-1.0, 1.0):
            Pp, fp = Pq.copy(), f.copy()
            if idx_arr == "P":
                Pp[50] = np.float32(float(Pq[50]) + sgn * h)
            else:
                fp[50] = np.float32(float(f[50]) + sgn * h)
            e.load_market_data(Pp, fp)
            vals.append((quantity(), float(Pp[50]), float(fp[50])))
        step = (vals[1][1] - vals[0][1]) if idx_arr == "P" else (vals[1][2] - vals[0][2])
        out.append(abs(vals[1][0] - vals[0][0]) / step)
    e.load_market_data(Pq, f)
    return out[0] * U_P + out[1] * U_F, out


def test_q1_bond_price_matches_the_readme(setup):
    """README.md:15-17: one Q1 run of 2^20 antithetic pairs printed P(0,0) = 1.000000 and P(0,10) = 0.876844 (exact under
    the README-era theta: 0.876777).  The GPU curve at the same sample size, 16 seeds: P(0,0) is exactly 1, the spread of
    P(0,10) over the seeds is the standard error BOTH single runs carry, and the README's value must sit within 3 of those
    of the GPU's mean (it is 1.3 standard errors from the exact value).  Not pinnable: the README's f(0,0) = 1.21 % -- the
    forward-difference formula at HEAD (market_data.cuh:121-126) gives 1.1987 % from the exact curve, i.e. 1.20 % to the
    README's two decimals, so that line was printed by an earlier formula."""
    hwmc, e, Pq, f = setup
    p10 = []
    for seed in range(16):
        Psum = e.simulate_zcb(N, 1000 + 7919 * seed)
        assert Psum[0] == 2.0 * N                                         # P(0,0) = 1.000000 exactly
        p10.append(Psum[100] / (2.0 * N))
    p10 = np.array(p10)
    se = float(p10.std(ddof=1))                                           # of ONE run of 2^20 pairs
    assert 3.0e-5 < se < 9.0e-5                                           # antithetic pairs: ~0.053 / sqrt(2^20) = 5.2e-5
    assert abs(float(p10.mean()) - 0.876777) < 3.0 * se / 4.0             # the GPU estimate against the exact value (16 runs)
    assert abs(README["P10"] - float(p10.mean())) < 3.0 * se * math.sqrt(1.0 + 1.0 / 16.0)
    assert abs(README["P10"] - 0.876777) == pytest.approx(U_P, abs=1e-6)  # the budget used below is that published deviation


def test_q1_checks_and_q2a_theta_recovery_against_the_readme(setup):
    """The two reference-held outputs that round 2 left aside (README.md:15-26):

        Q1  "P(0,0) = 1.000000, P(0,10) = 0.876844, f(0,0) = 1.21 %"  with the three range checks of 1_bond_pricing.cu:98-104
        Q2a "Max Error: 1.56e-03 (at T=0), Mean Error: 2.49e-04, SUCCESS (< 0.01)"      (2_option_pricing.cu:49-67: the
            errors are taken over the 11 printed maturities T = 0, 1, ..., 10, i.e. every SAVE_STRIDE-th of the 101)

    reproduced with the reference's own pipeline at its own sample size -- GPU curve of 2^20 antithetic pairs ->
    hw_average_and_forward (market_data.cuh:101-127) -> hw_recover_theta (2_option:14-35) -- under theta profile
    "readme", 16 seeds.

    What holds, and is asserted: the three range checks; SUCCESS; our errors stay INSIDE the README's (max <= 1.56e-3,
    mean <= 2.49e-4: they are upper budgets, the README's run was noisier than HEAD's estimator can be, see below);
    |f(0,0) - 1.21 %| within the 1.5e-4 the round-2 budget already allowed for that rate.

    What cannot be pinned, and why (asserted as facts about HEAD's formulas so that the statement stays true): with the
    forward-rate formula at HEAD, f(0,0) = -(ln P[1] - ln P[0]) / dT is an antithetic-pair estimate over [0, 0.1] whose
    Monte Carlo scatter at 2^20 pairs is ~1e-8 -- it prints 1.1987 % for every seed, 1.20 % to the README's two decimals,
    never 1.21 %; the theta error at T = 0 is then the deterministic 3.7e-4 of the one-sided difference, and the largest
    of the 11 errors sits at T = 10 (the other one-sided end, 5-7e-4, where P(0,10)'s sampling error enters).  The README's
    pair "1.21 %" and "1.56e-3 at T = 0" is self-consistent -- theta_rec(0) = (f[1] - f[0]) / dT + a f[0] gives an error of
    1.5e-3 when f[0] = 0.01215 -- so both lines were printed by an earlier, noisier f(0,0) estimate, not by the source at
    HEAD."""
    hwmc, e, Pq, f = setup
    p = e.params
    worst_max, worst_mean, at_T, f00, err0 = 0.0, 0.0, [], [], []
    idx = np.arange(0, p.n_mat, p.n_steps // (p.n_mat - 1))             # i = 0, SAVE_STRIDE, ... (2_option:49; common.cuh:29)
    assert len(idx) == 11
    for seed in range(16):
        Psum = e.simulate_zcb(N, 31337 + 7919 * seed)
        Pmc, fmc = e.compute_average_and_forward(Psum, 2 * N)
        # the three checks of 1_bond_pricing.cu:98-104, the reference's own acceptance test for Q1
        assert 0.99 < Pmc[0] < 1.01 and Pmc[0] == 1.0
        assert 0.3 < Pmc[100] < 0.9
        assert 0.01 < fmc[0] < 0.02
        rec, orig, Ts = e.recover_theta(fmc)
        err = np.abs(rec - orig)[idx]
        assert float(err.max()) < 0.01                                   # "Recovery: SUCCESS" (2_option:63)
        worst_max, worst_mean = max(worst_max, float(err.max())), max(worst_mean, float(err.mean()))
        at_T.append(float(Ts[idx][int(err.argmax())]))
        f00.append(float(fmc[0]))
        err0.append(float(err[0]))
    assert worst_max <= 1.56e-3 and worst_mean <= 2.49e-4, (worst_max, worst_mean)        # inside the README's figures
    assert abs(np.mean(f00) - 0.0121) < U_F                              # 1.21 % within the stated budget ...
    # ... and the facts behind "cannot be pinned to the printed digits":
    # (float32 P[1] next to 1 has a quantum of 6e-8, i.e. 6e-7 on f(0,0) and 6e-6 on the T = 0 error: that, not sampling, is the spread)
    assert max(f00) - min(f00) < 2e-6 and abs(100 * np.mean(f00) - 1.1987) < 2e-4 and all("%.2f" % (100 * v) == "1.20" for v in f00)
    assert max(err0) - min(err0) < 2e-5 and abs(np.mean(err0) - 3.7e-4) < 1.5e-5          # T = 0: deterministic 3.7e-4, not 1.56e-3
    assert set(at_T) == {10.0}                                           # the largest error sits at T = 10, not T = 0
    assert 4e-4 < worst_max < 9e-4 and 0.8e-4 < worst_mean < 1.6e-4     # measured 5.3-6.1e-4 / 1.1e-4 (oracle: the same)


def test_zbc_20_run_validation_matches_the_readme(setup):
    hwmc, e, Pq, f = setup
    P0S2 = README["P10"]                                              # 2_option:492: h_P[N_MAT-1]
    sums = e.simulate_ZBC_runs(N, 1234, n_runs=20, seed_stride=12345)
    res = [hwmc.cv_price(s, 2 * N, P0S2) for s in sums]
    price, ci = _t_ci([r["price"] for r in res])
    beta, beta_ci = _t_ci([r["beta"] for r in res])
    rho = float(np.mean([r["rho"] for r in res]))
    cv_pct = 100.0 * float(np.std([r["price"] for r in res], ddof=1)) / price

    budget, (dP, df) = _curve_budget(hwmc, e, Pq, f,
                                     lambda: hwmc.cv_price(e.simulate_ZBC_control_variate(N, 1234), 2 * N, P0S2)["price"])
    assert 0.3 < dP < 1.0 and 0.3 < df < 1.0                          # dZBC/dP50 ~ Phi(h) P(0,10)/P(0,5), dZBC/df50 ~ Phi(h) P(0,10) B
    combined = math.hypot(ci, README["zbc_ci"])
    # both CIs are ~8e-6; the unpublished P[50], f[50] dominate the budget (~1e-4 = 0.3 % of the price)
    assert ci < 2.0 * README["zbc_ci"]                                # same sample size, same estimator: same interval width
    assert abs(price - README["zbc"]) < combined + budget, (price, combined, budget)
    # ... and the price implied by the exact P[50], f[50] with the README's P(0,10) is predicted far more tightly by the
    # exact Hull-White value + the A-sensitivity: 0.03544557 + Phi(h) P/A dA (0.035492)
    assert abs(price - 0.035492) < 3.0 * combined, (price, combined)
    assert 0.5 * README["zbc_cv_pct"] < cv_pct < 1.6 * README["zbc_cv_pct"]       # sd of 20 runs: chi-square, 19 dof

    bbudget, _ = _curve_budget(hwmc, e, Pq, f,
                               lambda: hwmc.cv_price(e.simulate_ZBC_control_variate(N, 1234), 2 * N, P0S2)["beta"])
    assert beta_ci < 2.0 * README["beta_ci"]
    assert abs(beta - README["beta"]) < math.hypot(beta_ci, README["beta_ci"]) + bbudget, (beta, beta_ci, bbudget)
    assert abs(rho - README["rho"]) < 1.5e-3                          # printed to three decimals


def test_variance_reduction_of_the_optimal_control_variate(setup):
    """README.md:39 '+20.4 %' = 100 (1 - var(adjusted)/var(raw)) ACROSS runs (2_option:340-350).  With 20 runs that ratio of
    two sample variances scatters by tens of points, so it is taken over 200 batched runs here (40 ms of GPU): the
    antithetic pairs have already removed the linear part of the payoff, which is why the control variate only buys ~20 %
    and not the 1 - rho^2 = 55 % of plain sampling."""
    hwmc, e, Pq, f = setup
    sums = e.simulate_ZBC_runs(N, 777, n_runs=200, seed_stride=12345)
    res = [hwmc.cv_price(s, 2 * N, README["P10"]) for s in sums]
    adj = np.array([r["price"] for r in res])
    raw = np.array([r["mean_X"] for r in res])
    vr = 100.0 * (1.0 - adj.var(ddof=1) / raw.var(ddof=1))
    assert abs(vr - README["var_reduction_pct"]) < 12.0, vr
    # the run-to-run standard deviation of the price, pinned far better by 200 runs than by 20: the README's CV of 0.0496 %
    # is an sd of 1.76e-5 with a 16 % standard error of its own (19 degrees of freedom); 200 runs of the CPU oracle's
    # restatement at 2^13 pairs, scaled by sqrt(2^13 / 2^20), give 2.04e-5 +- 6 %
    sd = float(adj.std(ddof=1))
    assert 1.7e-5 < sd < 2.4e-5, sd
    assert abs(sd - 1.76e-5) < 3.0 * 0.16 * 1.76e-5                    # within 3 standard errors of the README's estimate
    # naive beta = 1 makes it WORSE (README.md:168: -42 %): the sign of that statement is checkable too
    naive = np.array([r["mean_X"] - 1.0 * (r["mean_Y"] - README["P10"]) for r in res])
    assert naive.var(ddof=1) > raw.var(ddof=1)


def test_vega_validation_matches_the_readme(setup):
    hwmc, e, Pq, f = setup
    runs = e.simulate_sensitivity_runs(N, 4242, n_runs=20, seed_stride=982451653) / N            # 3_sens:545, :261
    vega, ci = _t_ci(runs)
    cv_pct = 100.0 * float(np.std(runs, ddof=1)) / vega
    budget, _ = _curve_budget(hwmc, e, Pq, f, lambda: e.simulate_sensitivity(N, 4242) / N)
    assert ci < 2.0 * README["vega20_ci"]
    assert abs(vega - README["vega20"]) < math.hypot(ci, README["vega20_ci"]) + budget, (vega, ci, budget)
    assert 0.5 * README["vega_cv_pct"] < cv_pct < 1.6 * README["vega_cv_pct"]
    # single runs scatter by cv_pct: the README's own single pathwise run (0.229895) sits inside that scatter
    assert abs(README["vega_pw"] - vega) < 3.0 * cv_pct / 100.0 * vega
    # finite difference with common random numbers (3_sens:400-446), one 2^20-pair run: README 0.230316, "0.18 %" from pathwise
    fd = e.run_finite_difference(N, 4242 + 7)["vega_fd"]
    fd_se = 0.000089                                                   # the reference's own figure for it (3_sens:747)
    assert abs(fd - README["vega_fd"]) < 3.0 * math.sqrt(2.0) * fd_se + budget, fd
    assert abs(fd - vega) / vega < 0.0018 + (ci + 3.0 * fd_se) / vega, (fd, vega)   # README.md:50: PW and FD agree to 0.18 %
    # the diagnostic the README calls out as 127.74 % off (README.md:51): theta fixed, curve re-simulated
    rec = e.run_finite_difference_recalibrated(N, 99)["vega_fd_recalibrated"]
    assert abs(rec - 0.523552) < 0.03 and abs(rec / vega - 2.2774) < 0.15
