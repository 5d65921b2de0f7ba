"""CPU tests that pin the oracle (no GPU).

Pins, in order of strength:
  - Philox4x32-10: the Random123 known-answer vectors and rocRAND 4.2's own host implementation
    (bit-exact); Box-Muller normals against rocRAND's host rocrand_normal4 (float rounding).
  - constants / drift tables: the KAT values SURVEY.md section 2.3 derived from the reference's
    host code (include/common.cuh:60-110).
  - closed-form Hull-White answers and the reference's published outputs
    (README.md:15-56) as statistical targets.
The reference has no golden vectors and seeds from time(NULL): bit-level parity with a reference
run is unpinned (see oracle/hw_oracle.h).
"""
import math

import numpy as np
import pytest

# Random123 kat_vectors, philox4x32 10 rounds: (counter, key) -> output
R123_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_random123_kat(oracle_mod):
    o = oracle_mod.Oracle()
    for ctr, key, want in R123_KAT:
        assert o.philox(ctr, key) == want


def test_philox_matches_rocrand_bit_exact(oracle_mod, rocrand_vectors):
    o = oracle_mod.Oracle()
    for v in rocrand_vectors["vectors"]:
        got = []
        for b in range(3):
            got += o.uint4(v["seed"], v["subsequence"], b)
        assert got == v["words"], (v["seed"], v["subsequence"])


def test_normals_match_rocrand_host(oracle_mod, rocrand_vectors):
    # rocRAND's host path evaluates 2^-32 + x*2^-32 with a separate multiply and add where the
    # oracle (like hipcc's device code) fuses them: <= 1 ulp in the uniform, so compare loosely.
    o = oracle_mod.Oracle()
    worst = 0.0
    for v in rocrand_vectors["vectors"]:
        got = np.concatenate([o.normal4(v["seed"], v["subsequence"], b) for b in range(3)])
        worst = max(worst, float(np.max(np.abs(got - np.array(v["normals"], np.float32)))))
    assert worst < 2e-6, worst


def test_table_kats_from_survey(oracle_mod):
    """SURVEY.md section 2.3 KATs (f64-exact column, and f32 column to the 1-ulp-of-expf
    difference between NumPy's SIMD exp the survey used and glibc's expf used by real host code)."""
    o64 = oracle_mod.Oracle("head", precision="f64")
    d, sd = o64.tables()
    for i, want in ((0, 1.19471762e-4), (499, 1.88983624e-4), (500, 2.38853824e-4), (999, 2.88505153e-4)):
        assert d[i] == pytest.approx(want, rel=2e-8)
    assert sd[0] == pytest.approx(9.900581e-6, rel=2e-7)
    assert sd[499] == pytest.approx(9.949710e-4, rel=2e-7)
    o32 = oracle_mod.Oracle("head")
    d, sd = o32.tables()
    for i, want in ((0, 1.1947107e-4), (1, 1.1961038e-4), (499, 1.8898299e-4), (500, 2.3885237e-4), (999, 2.8850354e-4)):
        assert float(d[i]) == pytest.approx(want, rel=1e-5)
    assert float(sd[0]) == pytest.approx(9.89034e-6, rel=1e-6)
    assert float(sd[499]) == pytest.approx(9.949924e-4, rel=2e-5)
    k = o32.constants()
    assert float(k[0]) == pytest.approx(0.01) and float(k[1]) == pytest.approx(0.1)
    assert float(k[2]) == pytest.approx(0.9900499, rel=1e-6)
    assert float(k[3]) == pytest.approx(0.009950206, rel=1e-6)
    assert o32.steps_to(5.0) == 500
    assert oracle_mod.Oracle("head", n_steps=4000).steps_to(5.0) == 2000


def test_tables_f32_emulation_numpy(oracle_mod):
    """Independent float32 emulation of common.cuh:60-84 with correctly rounded libm calls."""
    o = oracle_mod.Oracle("head")
    d, sd = o.tables()
    f = np.float32
    dt = f(10.0) / f(1000)
    e = f(math.exp(-float(dt)))                 # expf is correctly rounded here
    c1 = (f(1) - e) / f(1)
    c2 = c1 / f(1)
    for i in (0, 1, 250, 499, 500, 999):
        s = f(i) * dt
        first = ((s + dt) - e * s) / f(1) - c2
        want = f(0.0014) * first + f(0.012) * c1 if s < f(5) else f(0.001) * first + f(0.019) * c1
        assert d[i] == want


def test_oracle_sums_do_not_depend_on_the_thread_count(oracle_mod, hwmc_mod):
    """Fixed chunks added in index order (oracle/hw_oracle.h, SUMMATION ORDER): 1, 3 and all threads agree bit for bit."""
    p = hwmc_mod.default_params("readme")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("readme")
    zs = [o.zbc_moments(P, f, 2500, 3, n_threads=t) for t in (1, 3, o.max_threads())]
    assert all(np.array_equal(zs[0], z) for z in zs)
    cs = [o.zcb_curve(1300, 5, n_threads=t) for t in (1, 3, o.max_threads())]
    assert all(np.array_equal(cs[0], c) for c in cs)
    vs = [o.vega_pathwise(P, f, 2500, 4, n_threads=t) for t in (1, 3, o.max_threads())]
    assert vs[0] == vs[1] == vs[2]


def test_golden_matches_oracle(oracle_mod, hwmc_mod, golden):
    """The committed fixture is what the oracle produces today (guards accidental drift)."""
    for case in golden["cases"]:
        if case["n"] > 1024:
            continue
        p = hwmc_mod.default_params(case["profile"])
        P, f = hwmc_mod.analytic_curve(p)
        o = oracle_mod.Oracle(case["profile"], precision=case["precision"])
        s = o.zbc_moments(P, f, case["n"], case["seed"], path_offset=case["path_offset"])
        np.testing.assert_allclose(s, case["zbc_sums"], rtol=1e-12)
        v = o.vega_pathwise(P, f, case["n"], case["seed"], path_offset=case["path_offset"])
        assert v == pytest.approx(case["vega_sum"], rel=1e-12)


@pytest.mark.parametrize("profile,zbc,beta,rho,vega,P10", [
    ("head", 0.02525551, 0.135609, 0.6330, 0.240117, 0.859387),
    ("readme", 0.03544557, 0.166309, 0.6738, 0.230244, 0.876777),
])
def test_oracle_vs_closed_form(oracle_mod, hwmc_mod, profile, zbc, beta, rho, vega, P10):
    """Exact Hull-White answers: price within 4 SE, beta*/rho/Vega within
    their sampling error, MC curve within 4 SE."""
    p = hwmc_mod.default_params(profile)
    P, f = hwmc_mod.analytic_curve(p)
    assert float(P[100]) == pytest.approx(P10, abs=2e-6)
    o = oracle_mod.Oracle(profile)
    n = 1 << 17
    s = o.zbc_moments(P, f, n, 20240613)
    r = o.cv_price(s, 2 * n, float(P[-1]))
    se = math.sqrt(max(r["var_X"] - r["cov"] ** 2 / r["var_Y"], 0) / (2 * n))
    assert abs(r["price"] - zbc) < 4 * se
    assert r["beta"] == pytest.approx(beta, abs=2e-3)
    assert r["rho"] == pytest.approx(rho, abs=5e-3)
    v = o.vega_pathwise(P, f, n, 5) / n
    assert v == pytest.approx(vega, abs=4e-3)
    c = o.zcb_curve(1 << 13, 11)
    Pmc, fmc = o.average_and_forward(c, 2 << 13)
    assert Pmc[0] == 1.0
    assert float(Pmc[100]) == pytest.approx(P10, abs=2.5e-3)
    assert float(np.max(np.abs(Pmc - P))) < 3e-3


def test_readme_published_numbers_statistically(oracle_mod, hwmc_mod):
    """README.md:15-17,36-56 under theta profile 'readme': P(0,10)=0.876844, beta*=0.166447,
    rho=0.673, ZBC=0.035492 (fed by the README's own MC P(0,10)), Vega PW 0.229895 / FD 0.230316."""
    p = hwmc_mod.default_params("readme")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("readme")
    n = 1 << 17
    Pq = P.copy()
    Pq[100] = np.float32(0.876844)                  # the published Q1 estimate the README's Q2 consumed
    s = o.zbc_moments(Pq, f, n, 31337)
    r = o.cv_price(s, 2 * n, 0.876844)
    se = math.sqrt(max(r["var_X"] - r["cov"] ** 2 / r["var_Y"], 0) / (2 * n))
    assert abs(r["price"] - 0.03549203) < 4 * se + 3 * 0.00000825
    assert r["beta"] == pytest.approx(0.166447, abs=2.5e-3)
    assert r["rho"] == pytest.approx(0.673, abs=5e-3)
    v = o.vega_pathwise(P, f, n, 8) / n
    assert v == pytest.approx(0.229895, abs=4e-3)
    eps, s0 = 1e-3, 0.1
    pr = []
    for sg in (s0 - eps, s0 + eps):
        sm = o.zbc_moments(P, f, n, 9, sigma=sg, drift=o.shifted_drift(sg, s0))
        pr.append(o.cv_price(sm, 2 * n, float(P[-1]))["price"])
    fd = (pr[1] - pr[0]) / (2 * eps)
    assert fd == pytest.approx(0.230316, abs=4e-3)


def test_theta_recovery_and_statistics(oracle_mod, hwmc_mod):
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("head")
    rec, orig, Ts = o.recover_theta(f)
    err = np.abs(rec - orig)[::10]
    assert float(err.max()) < 0.01                 # 2_option:65 "SUCCESS" threshold
    assert float(Ts[100]) == pytest.approx(10.0)
    st = o.run_statistics(np.linspace(0.0354, 0.0356, 20))
    x = np.linspace(0.0354, 0.0356, 20)
    assert st["mean"] == pytest.approx(x.mean(), rel=1e-6)
    assert st["margin"] == pytest.approx(2.093 * x.std(ddof=1) / math.sqrt(20), rel=1e-4)


def test_oracle_edge_cases(oracle_mod, hwmc_mod):
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("head")
    assert np.all(o.zbc_moments(P, f, 0, 1) == 0.0)          # empty
    whole = o.zbc_moments(P, f, 1000, 3)
    parts = o.zbc_moments(P, f, 333, 3) + o.zbc_moments(P, f, 667, 3, path_offset=333)
    np.testing.assert_allclose(whole, parts, rtol=1e-13)     # sharding by path_offset
    one = o.zbc_moments(P, f, 1000, 3, n_threads=1)
    np.testing.assert_allclose(whole, one, rtol=1e-13)       # thread count does not matter
