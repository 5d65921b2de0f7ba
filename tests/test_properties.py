"""Property tests (hypothesis) of the host-side pieces of the path: the control-variate algebra behind beta*,
the shard partition, the statistics of the 20-run harness and the oracle's counter-based RNG.  CPU only."""
import math

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


@settings(max_examples=60, deadline=None)
@given(st.integers(min_value=4, max_value=200), st.integers(min_value=0, max_value=2 ** 32 - 1),
       st.floats(min_value=0.1, max_value=2.0))
def test_cv_price_is_the_regression_adjusted_mean(hwmc_mod, n, seed, P0S2):
    """hw_cv_price on the five sums of ANY sample equals the textbook estimator (2_option:154-175): means, population
    variance/covariance, beta = cov/varY, price = meanX - beta (meanY - P0S2); it is invariant under a shift of Y
    together with P0S2 and scales linearly in X."""
    rng = np.random.default_rng(seed)
    y = rng.normal(0.9, 0.05, n)
    x = np.maximum(0.3 * y - 0.25 + rng.normal(0, 0.01, n), 0.0)
    sums = [x.sum(), y.sum(), (x * x).sum(), (y * y).sum(), (x * y).sum()]
    r = hwmc_mod.cv_price(sums, n, P0S2)
    varY = (y * y).mean() - y.mean() ** 2
    cov = (x * y).mean() - x.mean() * y.mean()
    assert r["mean_X"] == pytest.approx(x.mean(), rel=1e-12)
    assert r["var_Y"] == pytest.approx(varY, rel=1e-9, abs=1e-15)
    assert r["beta"] == pytest.approx(cov / varY, rel=1e-8, abs=1e-12)
    assert r["price"] == pytest.approx(x.mean() - cov / varY * (y.mean() - P0S2), rel=1e-9, abs=1e-12)
    assert -1.0 - 1e-9 <= r["rho"] <= 1.0 + 1e-9
    x3 = 3.0 * x
    r3 = hwmc_mod.cv_price([x3.sum(), y.sum(), (x3 * x3).sum(), (y * y).sum(), (x3 * y).sum()], n, P0S2)
    assert r3["price"] == pytest.approx(3.0 * r["price"], rel=1e-8, abs=1e-10)
    ys = y + 0.125
    rs = hwmc_mod.cv_price([x.sum(), ys.sum(), (x * x).sum(), (ys * ys).sum(), (x * ys).sum()], n, P0S2 + 0.125)
    assert rs["price"] == pytest.approx(r["price"], rel=1e-7, abs=1e-10)


@settings(max_examples=200, deadline=None)
@given(st.integers(min_value=0, max_value=2 ** 45), st.integers(min_value=1, max_value=64))
def test_shard_is_a_balanced_contiguous_partition(hwmc_mod, n, world):
    end, sizes = 0, []
    for rank in range(world):
        off, cnt = hwmc_mod.parallel.shard(n, rank, world)
        assert off == end and cnt >= 0
        end = off + cnt
        sizes.append(cnt)
    assert end == n and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


@settings(max_examples=50, deadline=None)
@given(st.lists(st.floats(min_value=0.01, max_value=1.0), min_size=20, max_size=20))
def test_run_statistics_is_the_t_interval_of_the_reference(hwmc_mod, xs):
    s = hwmc_mod.run_statistics(xs)
    x = np.array(xs)
    assert s["mean"] == pytest.approx(x.mean(), rel=1e-12)
    assert s["sd"] == pytest.approx(x.std(ddof=1), rel=1e-9, abs=1e-15)
    assert s["ci95_width"] == pytest.approx(2 * 2.093 * x.std(ddof=1) / math.sqrt(20), rel=1e-9, abs=1e-15)   # 2_option:318-320
    assert s["ci_lo"] <= s["mean"] <= s["ci_hi"]


@settings(max_examples=40, deadline=None)
@given(st.integers(min_value=0, max_value=2 ** 64 - 1), st.integers(min_value=0, max_value=2 ** 64 - 1),
       st.integers(min_value=0, max_value=2 ** 32 - 1))
def test_oracle_philox_is_a_pure_function_of_seed_subsequence_block(oracle_mod, seed, sub, block):
    """Counter-based: the four words of (seed, subsequence, block) do not depend on call order, differ between
    neighbouring blocks and subsequences, and the normals built from them are finite and inside Box-Muller's range."""
    o = oracle_mod.Oracle()
    a = o.uint4(seed, sub, block)
    assert a == o.uint4(seed, sub, block)
    assert a != o.uint4(seed, sub, (block + 1) % 2 ** 32) and a != o.uint4(seed, (sub + 1) % 2 ** 64, block)
    g = np.array(o.normal4(seed, sub, block), np.float64)
    assert np.all(np.isfinite(g)) and np.all(np.abs(g) <= math.sqrt(2 * 32 * math.log(2)) + 1e-5)


def test_usable_cores_never_exceeds_the_openmp_default(oracle_mod):
    n, how = oracle_mod.usable_cores(1000)
    assert 1 <= n <= 1000 and how in ("omp_get_max_threads", "sched_getaffinity", "cgroup cpu quota")
    assert oracle_mod.usable_cores(1) == (1, "omp_get_max_threads")


@pytest.mark.parametrize("profile", ["head", "readme"])
def test_analytic_curve_against_numerical_quadrature(hwmc_mod, profile):
    """hw_analytic_curve (the synthetic market input of bench.py and of most parity tests) against an independent
    evaluation: for dr = (theta(t) - a r) dt + sigma dW,  int_0^T r  is Gaussian with
    mean r0 B(0,T) + int_0^T theta(u) B(u,T) du  and variance sigma^2 int_0^T B(u,T)^2 du, B(u,T) = (1 - e^{-a(T-u)})/a,
    so P(0,T) = exp(-mean + variance/2) and f(0,T) = r0 e^{-aT} + int_0^T theta(u) e^{-a(T-u)} du - sigma^2 B(0,T)^2 / 2.
    The integrals are done with scipy.integrate.quad on the piecewise-linear theta."""
    from scipy.integrate import quad
    p = hwmc_mod.default_params(profile)
    P, f = hwmc_mod.analytic_curve(p)
    a, s, r0 = p.a, p.sigma, p.r0

    def theta(u):
        return p.th_a1 + p.th_b1 * u if u < p.th_break else p.th_a2 + p.th_b2 * u

    def B(u, T):
        return (1.0 - math.exp(-a * (T - u))) / a

    for j in (1, 17, 50, 51, 77, 100):
        T = j * p.T_final / (p.n_mat - 1)
        pts = [p.th_break] if T > p.th_break else None
        mean = r0 * B(0.0, T) + quad(lambda u: theta(u) * B(u, T), 0.0, T, points=pts, epsabs=1e-13, epsrel=1e-13)[0]
        var = s * s * quad(lambda u: B(u, T) ** 2, 0.0, T, epsabs=1e-14, epsrel=1e-13)[0]
        fwd = r0 * math.exp(-a * T) + quad(lambda u: theta(u) * math.exp(-a * (T - u)), 0.0, T, points=pts,
                                           epsabs=1e-13, epsrel=1e-13)[0] - 0.5 * s * s * B(0.0, T) ** 2
        assert float(P[j]) == pytest.approx(math.exp(-mean + 0.5 * var), rel=2e-7)      # float32 storage
        assert float(f[j]) == pytest.approx(fwd, rel=5e-7, abs=1e-8)
    assert P[0] == 1.0 and float(f[0]) == pytest.approx(r0, rel=1e-7)


def test_closed_form_targets_of_the_parity_tests_are_derived_not_magic(hwmc_mod):
    """The constants the statistical parity tests compare against (tests/test_oracle.py, tests/test_gpu_parity.py:
    ZBC 0.03544557 and Vega 0.230244 under the readme profile) are Jamshidian's bond-option formula on the exact
    curve and its derivative in sigma with the market curves held fixed -- the quantity both the pathwise and the
    shifted-drift finite-difference estimators of the reference target (3_sens:15-19, 374-398)."""
    from scipy.stats import norm
    p = hwmc_mod.default_params("readme")
    P, _ = hwmc_mod.analytic_curve(p)
    a, S1, S2, K = p.a, 5.0, 10.0, float(np.exp(np.float32(-0.1)))
    P1, P2 = float(P[50]), float(P[100])
    B = (1.0 - math.exp(-a * (S2 - S1))) / a

    def price(sig):
        sp = sig * math.sqrt((1.0 - math.exp(-2.0 * a * S1)) / (2.0 * a)) * B
        h = math.log(P2 / (K * P1)) / sp + 0.5 * sp
        return P2 * norm.cdf(h) - K * P1 * norm.cdf(h - sp)

    assert price(p.sigma) == pytest.approx(0.03544557, abs=5e-8)
    assert (price(p.sigma + 1e-5) - price(p.sigma - 1e-5)) / 2e-5 == pytest.approx(0.230244, abs=2e-6)
