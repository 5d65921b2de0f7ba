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


# ---- a SECOND, independent restatement of the per-path bodies (NumPy float32, written from the reference source, not from
# ---- oracle/): a transcription mistake in the C oracle would have to be made twice, in two languages, to go unnoticed -------
def _np_normals(o, seed, n_pairs, n_steps):
    """g[pair, step] from the pinned RNG layer (Philox + Box-Muller: tests above): step i uses component i & 3 of block i >> 2."""
    g = np.zeros((n_pairs, n_steps), np.float32)
    for pid in range(n_pairs):
        for b in range((n_steps + 3) // 4):
            g[pid, 4 * b:4 * b + 4] = o.normal4(seed, pid, b)[: n_steps - 4 * b]
    return g


def _np_fma(a, b, c):
    """__fmaf_rn(a, b, c) for float32 arrays: the product of two floats is exact in double; one rounding to double, one to float."""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)


def _np_step(r, integral, drift, sig_g, e, dt):
    """evolve_hull_white_step, common.cuh:237-244."""
    f = np.float32
    r_next = _np_fma(r, e, (f(drift) + sig_g).astype(np.float32))
    integral = (integral + (f(0.5) * (r + r_next)).astype(np.float32) * f(dt)).astype(np.float32)
    return r_next, integral


def _np_interp(data, T, spacing):
    """interpolate, common.cuh:187-196."""
    f = np.float32
    idx = int(f(T) / f(spacing))
    if idx >= len(data) - 1:
        return f(data[-1])
    t0 = f(idx) * f(spacing)
    alpha = (f(T) - t0) / f(spacing)
    return f(f(data[idx]) * (f(1) - alpha) + f(data[idx + 1]) * alpha)


@pytest.mark.parametrize("profile", ["head", "readme"])
def test_second_restatement_of_the_zbc_body_in_numpy(oracle_mod, hwmc_mod, profile):
    """simulate_ZBC_control_variate (common.cuh:286-362) with B_func / compute_A_HW / compute_P_HW (common.cuh:180-225),
    statement by statement in NumPy float32 on 192 pairs, against the C oracle's five sums: 2e-6 relative (NumPy's float32
    exp is not glibc's expf: up to an ulp per call)."""
    f = np.float32
    p = hwmc_mod.default_params(profile)
    P, fw = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle(profile)
    n, seed, S1, S2, K = 192, 20240613, f(5.0), f(10.0), f(np.exp(f(-0.1)))
    dt, _, e, sig_st = (f(x) for x in o.constants()[:4])
    drift, _ = o.tables()
    a, sigma, r0, spacing = f(1.0), f(0.1), f(0.012), f(10.0) / f(100)
    n_steps = int(S1 / dt)                                               # common.cuh:322
    assert n_steps == 500
    g = _np_normals(o, seed, n, n_steps)
    r1 = np.full(n, r0, np.float32)
    r2 = r1.copy()
    i1 = np.zeros(n, np.float32)
    i2 = i1.copy()
    for i in range(n_steps):                                             # common.cuh:325-332
        sig_g = (sig_st * g[:, i]).astype(np.float32)
        r1, i1 = _np_step(r1, i1, drift[i], sig_g, e, dt)
        r2, i2 = _np_step(r2, i2, drift[i], (-sig_g).astype(np.float32), e, dt)
    B = (f(1) - np.exp(-a * (S2 - S1), dtype=np.float32)) / a           # B_func
    P0T, P0t, f0t = _np_interp(P, S2, spacing), _np_interp(P, S1, spacing), _np_interp(fw, S1, spacing)
    term3 = (sigma * sigma / (f(4) * a)) * (f(1) - np.exp(f(-2) * a * S1, dtype=np.float32)) * B * B
    A = (P0T / P0t) * np.exp(f(B * f0t - term3), dtype=np.float32)      # compute_A_HW
    P1 = (A * np.exp((-B * r1).astype(np.float32))).astype(np.float32)  # compute_P_HW
    P2 = (A * np.exp((-B * r2).astype(np.float32))).astype(np.float32)
    d1, d2 = np.exp(-i1).astype(np.float32), np.exp(-i2).astype(np.float32)
    c1, c2 = d1 * P1, d2 * P2
    x1, x2 = d1 * np.maximum(P1 - K, f(0)), d2 * np.maximum(P2 - K, f(0))
    sums = np.array([np.sum((x1 + x2).astype(np.float64)), np.sum((c1 + c2).astype(np.float64)),
                     np.sum((x1 * x1 + x2 * x2).astype(np.float64)), np.sum((c1 * c1 + c2 * c2).astype(np.float64)),
                     np.sum((x1 * c1 + x2 * c2).astype(np.float64))])
    np.testing.assert_allclose(sums, o.zbc_moments(P, fw, n, seed), rtol=2e-6)
    assert sums[0] > 0 and np.count_nonzero(x1 + x2) > n // 4           # the option is in the money on a fair share of the paths


def test_second_restatement_of_the_curve_body_in_numpy(oracle_mod):
    """simulate_zcb (market_data.cuh:35-67): both antithetic legs, exp(-integral) summed at every SAVE_STRIDE-th step,
    P_sum[0] = 2 N, in NumPy float32 on 96 pairs against the C oracle's P_sum[101]."""
    f = np.float32
    o = oracle_mod.Oracle("head")
    n, seed, n_steps, stride = 96, 777, 1000, 10
    dt, _, e, sig_st = (f(x) for x in o.constants()[:4])
    drift, _ = o.tables()
    g = _np_normals(o, seed, n, n_steps)
    r1 = np.full(n, f(0.012), np.float32)
    r2 = r1.copy()
    i1 = np.zeros(n, np.float32)
    i2 = i1.copy()
    want = np.zeros(101)
    want[0] = 2.0 * n
    for i in range(1, n_steps + 1):
        sig_g = (sig_st * g[:, i - 1]).astype(np.float32)
        r1, i1 = _np_step(r1, i1, drift[i - 1], sig_g, e, dt)
        r2, i2 = _np_step(r2, i2, drift[i - 1], (-sig_g).astype(np.float32), e, dt)
        if i % stride == 0:
            want[i // stride] = np.sum((np.exp(-i1).astype(np.float32) + np.exp(-i2).astype(np.float32)).astype(np.float64))
    np.testing.assert_allclose(o.zcb_curve(n, seed), want, rtol=2e-6)


def test_second_restatement_of_the_vega_body_in_numpy(oracle_mod, hwmc_mod):
    """simulate_sensitivity + compute_dP_dsigma (3_sensitivity_analysis.cu:15-19, 36-80): r and d_sigma r driven by the same
    normal, one path per draw, in NumPy float32 on 160 paths against the C oracle's pathwise-Vega sum."""
    f = np.float32
    p = hwmc_mod.default_params("readme")
    P, fw = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("readme")
    n, seed, S1, S2, K = 160, 4242, f(5.0), f(10.0), f(np.exp(f(-0.1)))
    dt, _, e, sig_st = (f(x) for x in o.constants()[:4])
    drift, sdrift = o.tables()
    a, sigma, spacing = f(1.0), f(0.1), f(10.0) / f(100)
    n_steps = int(S1 / dt)
    g = _np_normals(o, seed, n, n_steps)
    r = np.full(n, f(0.012), np.float32)
    dr = np.zeros(n, np.float32)
    ir = np.zeros(n, np.float32)
    idr = np.zeros(n, np.float32)
    for i in range(1, n_steps + 1):                                      # 3_sens:47-61
        r, ir = _np_step(r, ir, drift[i - 1], (sig_st * g[:, i - 1]).astype(np.float32), e, dt)
        dr, idr = _np_step(dr, idr, sdrift[i - 1], ((sig_st / sigma) * g[:, i - 1]).astype(np.float32), e, dt)
    B = (f(1) - np.exp(-a * (S2 - S1), dtype=np.float32)) / a
    P0T, P0t, f0t = _np_interp(P, S2, spacing), _np_interp(P, S1, spacing), _np_interp(fw, S1, spacing)
    term3 = (sigma * sigma / (f(4) * a)) * (f(1) - np.exp(f(-2) * a * S1, dtype=np.float32)) * B * B
    A = (P0T / P0t) * np.exp(f(B * f0t - term3), dtype=np.float32)
    Pv = (A * np.exp((-B * r).astype(np.float32))).astype(np.float32)   # 3_sens:65
    disc = np.exp(-ir).astype(np.float32)                               # 3_sens:68
    one_minus = f(1) - np.exp(f(-2) * a * S1, dtype=np.float32)
    dP = (-Pv * B * (sigma / (f(2) * a) * one_minus * B + dr)).astype(np.float32)      # compute_dP_dsigma
    term1 = np.where(Pv > K, dP * disc, f(0)).astype(np.float32)        # 3_sens:72-74
    term2 = (idr * disc * np.maximum(Pv - K, f(0))).astype(np.float32)  # 3_sens:77-78
    got = float(np.sum((term1 - term2).astype(np.float64)))
    assert got == pytest.approx(o.vega_pathwise(P, fw, n, seed), rel=3e-6)
    assert 0.1 < got / n < 0.4                                           # a Vega near 0.23 on 160 paths


def test_vega_terms_scale_is_consistent_with_the_vega_sum(oracle_mod, hwmc_mod):
    """ora_vega_pathwise_terms (round 4: the scale of the Vega sum's signed terms, used by the GPU parity bounds) runs the
    SAME per-path body as ora_vega_pathwise: its first output is that sum bit for bit (golden fixture included), the two
    absolute sums bound it, float and double agree, and shards add up."""
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    for prec, tol in (("f32", 2e-5), ("f64", 1e-12)):
        o = oracle_mod.Oracle("head", precision=prec)
        for n, seed, off in ((1024, 1234, 0), (1000, 777, 123456789012)):
            s, a1, a2 = o.vega_pathwise_terms(P, f, n, seed, path_offset=off)
            assert s == o.vega_pathwise(P, f, n, seed, path_offset=off)
            assert a1 > 0 and a2 > 0 and abs(s) <= (a1 + a2) * (1 + 1e-12)
            h = n // 2
            parts = [o.vega_pathwise_terms(P, f, h, seed, path_offset=off), o.vega_pathwise_terms(P, f, n - h, seed, path_offset=off + h)]
            for k in range(3):
                assert parts[0][k] + parts[1][k] == pytest.approx((s, a1, a2)[k], rel=1e-12)
            assert o.vega_pathwise_terms(P, f, n, seed, path_offset=off, n_threads=1) == pytest.approx((s, a1, a2), rel=1e-13)
    s32 = oracle_mod.Oracle("head", precision="f32").vega_pathwise_terms(P, f, 4096, 99)
    s64 = oracle_mod.Oracle("head", precision="f64").vega_pathwise_terms(P, f, 4096, 99)
    assert s32 == pytest.approx(s64, rel=2e-5)
