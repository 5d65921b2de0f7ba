"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same Philox stream, against the committed golden vectors, and --
at BASELINE.json's full sizes -- through size-independent properties.

Tolerances (north_star: prices, beta*, Vega within 1e-5 relative; integer path counts exact):
  REL_F32 = 1e-5  moments / price / beta* of the float32 engine vs the float32 oracle.  The GPU
                  uses the hardware v_log/v_sqrt/v_sin/v_cos/v_exp approximations and the
                  superposed stepper (different rounding order), the oracle uses libm and the
                  reference's per-step arithmetic; observed differences are ~1e-7..1e-6.
  REL_F64 = 2e-6  float64 engine vs float64 oracle: both consume the SAME float32 Box-Muller
                  normals, whose GPU/libm transcendental differences (~1e-7 per draw) set the floor.
  Philox words: bit-exact.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_F32 = 1e-5
REL_F64 = 2e-6


@pytest.fixture(scope="module")
def setup(hwmc_mod, oracle_mod):
    engines = {}

    def get(profile="head", precision="f32", n_steps=1000, table_mode="f64"):
        key = (profile, precision, n_steps, table_mode)
        if key not in engines:
            p = hwmc_mod.default_params(profile, n_steps=n_steps, precision=precision, table_mode=table_mode)
            e = hwmc_mod.HullWhiteEngine(p, device=0)
            P, f = hwmc_mod.analytic_curve(p)
            e.load_market_data(P, f)
            o = oracle_mod.Oracle(profile, n_steps=n_steps, precision=precision)
            engines[key] = (e, o, P, f)
        return engines[key]

    yield get
    for e, _, _, _ in engines.values():
        e.close()


def _price(hwmc_mod, sums, n_pairs, P):
    return hwmc_mod.cv_price(sums, 2 * n_pairs, float(P[-1]))


# ---- RNG --------------------------------------------------------------------------------------
def test_device_philox_bit_exact_vs_rocrand_and_oracle(setup, rocrand_vectors):
    e, o, _, _ = setup()
    for v in rocrand_vectors["vectors"]:
        got = e.debug_raw(v["seed"], v["subsequence"], 3).reshape(-1).tolist()
        assert got == v["words"], (v["seed"], v["subsequence"])
    # deep into a path (block 124 = steps 496..499) and 64-bit ids
    for seed, sub in ((1234, 0), (20240613, 123456789012), (2 ** 63 + 5, 2 ** 40 + 3)):
        raw = e.debug_raw(seed, sub, 125)
        for b in (0, 1, 63, 124):
            assert raw[b].tolist() == o.uint4(seed, sub, b)


def test_device_normals_vs_oracle_and_rocrand(setup, rocrand_vectors, golden):
    e, o, _, _ = setup()
    worst = 0.0
    for v in rocrand_vectors["vectors"]:
        got = e.debug_normals(v["seed"], v["subsequence"], 3).reshape(-1)
        worst = max(worst, float(np.max(np.abs(got - np.array(v["normals"], np.float32)))))
    for g in golden["normals"]:
        got = e.debug_normals(g["seed"], g["subsequence"], g["block"] + 1)[g["block"]]
        worst = max(worst, float(np.max(np.abs(got - np.array(g["g"], np.float32)))))
    # hardware sin/cos/log/sqrt vs libm: absolute error of a N(0,1) draw
    assert worst < 2e-5, worst
    # moments of 2^16 device normals
    z = e.debug_normals(99, 7, 1 << 14).reshape(-1).astype(np.float64)
    assert abs(z.mean()) < 4 / math.sqrt(z.size)
    assert abs(z.var() - 1.0) < 4 * math.sqrt(2.0 / z.size)


def test_device_exp2_of_a_double(setup):
    """The double engine's exponential is v_exp_f32 on the float-rounded argument plus a first-order correction
    (hw_exp2(double), csrc/hw_kernels.hpp; include/hw_mc.h HW_F64): against libm's exp2 over the arguments the path
    produces (|x| = B r log2e, int r log2e: O(1); checked out to +-40) the relative error must stay below 1.2e-7
    (v_exp_f32: 1 ulp of float) and its MEAN signed error below 1e-8 -- a biased exponential shifts every discount factor
    the same way, which no amount of paths averages out (measured, tools/exp2_probe.py over 4 M arguments: max 8.4e-8,
    sd 2.6e-8, mean -2.3e-9, i.e. three orders below the engine's 2e-6 parity bar).  The reference holds nothing that pins a per-path exp error (its
    kernels are float32, common.cuh:337-343), so this is a property test, not a parity pin."""
    e, _, _, _ = setup("head", "f64")
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-3.0, 3.0, 1 << 18), rng.normal(0.0, 0.2, 1 << 18), rng.uniform(-40.0, 40.0, 1 << 16),
                        np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1e-30, -1e-30, 2.0 ** -30, 100.0, -100.0])])
    got = e.debug_exp2(x)
    rel = got / np.exp2(x) - 1.0
    assert float(np.max(np.abs(rel))) < 1.2e-7, float(np.max(np.abs(rel)))
    core = rel[: 1 << 19]
    assert abs(float(core.mean())) < 1e-8, (core.mean(), core.std())
    assert got[-10] == 1.0 and got[-9] == 1.0 and got[-8] == 2.0 and got[-7] == 0.5          # exact at 0, -0, 1, -1


# ---- ZBC: the north-star kernel -----------------------------------------------------------------
@pytest.mark.parametrize("profile", ["head", "readme"])
@pytest.mark.parametrize("stepper", ["superposed", "reference"])
def test_zbc_moments_vs_oracle_f32(setup, hwmc_mod, profile, stepper):
    e, o, P, f = setup(profile)
    st = hwmc_mod.HW_STEPPER_SUPERPOSED if stepper == "superposed" else hwmc_mod.HW_STEPPER_REFERENCE
    for n, seed in ((1 << 10, 1234), (1 << 14, 20240613), (1 << 16, 42)):
        got = e.simulate_ZBC_control_variate(n, seed, stepper=st)
        want = o.zbc_moments(P, f, n, seed)
        np.testing.assert_allclose(got, want, rtol=REL_F32)
        pg, pw = _price(hwmc_mod, got, n, P), _price(hwmc_mod, want, n, P)
        assert pg["price"] == pytest.approx(pw["price"], rel=REL_F32)
        assert pg["beta"] == pytest.approx(pw["beta"], rel=REL_F32)


def test_zbc_moments_vs_golden_fixture(setup, hwmc_mod, golden):
    for case in golden["cases"]:
        e, _, P, f = setup(case["profile"], case["precision"])
        got = e.simulate_ZBC_control_variate(case["n"], case["seed"], path_offset=case["path_offset"])
        tol = REL_F32 if case["precision"] == "f32" else REL_F64
        np.testing.assert_allclose(got, case["zbc_sums"], rtol=tol)


def test_zbc_f64_engine_vs_oracle_f64(setup, hwmc_mod):
    e, o, P, f = setup("head", "f64")
    n = 1 << 14
    got = e.simulate_ZBC_control_variate(n, 7)
    want = o.zbc_moments(P, f, n, 7)
    np.testing.assert_allclose(got, want, rtol=REL_F64)


def test_zbc_reference_tables_mode(setup, hwmc_mod):
    """table_mode=ref_f32 (the reference's float32 host tables) against the float32 oracle."""
    e, o, P, f = setup("head", "f32", 1000, "ref_f32")
    n = 1 << 14
    for st in (hwmc_mod.HW_STEPPER_SUPERPOSED, hwmc_mod.HW_STEPPER_REFERENCE):
        got = e.simulate_ZBC_control_variate(n, 5, stepper=st)
        np.testing.assert_allclose(got, o.zbc_moments(P, f, n, 5), rtol=REL_F32)
    d, sd = e.drift_tables()
    do, sdo = o.tables()
    np.testing.assert_array_equal(d.astype(np.float32), do)       # same float32 arithmetic
    np.testing.assert_allclose(sd, sdo, rtol=2e-5)                # coshf difference: clang vs glibc paths agree to ulps


def test_zbc_lds_table_variant_and_launch_knobs(setup, hwmc_mod):
    e, o, P, f = setup("head")
    n = 5000
    want = o.zbc_moments(P, f, n, 77)
    try:
        for block, cap, lds, ilp in ((256, 0, 1, 1), (64, 1, 0, 1), (1024, 1, 1, 2), (512, 2, 0, 2), (128, 0, 0, 2)):
            e.set_launch(block, cap, lds, ilp)
            for st in (hwmc_mod.HW_STEPPER_SUPERPOSED, hwmc_mod.HW_STEPPER_REFERENCE):
                got = e.simulate_ZBC_control_variate(n, 77, stepper=st)
                np.testing.assert_allclose(got, want, rtol=REL_F32)
            # the other two kernels under the same geometry (wave count per workgroup changes the LDS rows)
            np.testing.assert_allclose(e.simulate_zcb(700, 5), o.zcb_curve(700, 5), rtol=REL_F32)
            assert e.simulate_sensitivity(n, 9) == pytest.approx(o.vega_pathwise(P, f, n, 9), rel=REL_F32)
    finally:
        e.set_launch(-1, 0, 0, -1)


def test_two_pairs_in_flight_interleaved_blocks(setup, hwmc_mod):
    """pairs_in_flight = 2 (the engine's automatic choice from two chip-fills of pairs up): the Philox blocks of a
    thread's two pairs run in lockstep (philox_block_prio_x2).  Per pair the arithmetic is the one-pair loop's, so the
    sums must equal the one-pair launch to 1e-13 (only the grouping of the double additions differs) and the oracle to
    1e-5 -- ragged counts, ids straddling a multiple of 2^32 (the per-lane round-2 multiply), 1..3 tail steps, the Vega
    kernel, batched runs, and the automatic switch at two chip-fills (2^20 pairs on 256 CUs)."""
    e, o, P, f = setup("head")
    try:
        for n, off in ((5000, 0), (70_001, 123), (4097, (1 << 32) - 2000), (300, (5 << 32) - 150)):
            want = o.zbc_moments(P, f, n, 77, path_offset=off)
            e.set_launch(pairs_in_flight=1)
            one = e.simulate_ZBC_control_variate(n, 77, path_offset=off)
            e.set_launch(pairs_in_flight=2)
            two = e.simulate_ZBC_control_variate(n, 77, path_offset=off)
            np.testing.assert_allclose(two, one, rtol=1e-13, err_msg="n=%d off=%d" % (n, off))
            np.testing.assert_allclose(two, want, rtol=REL_F32)
            v1 = e.simulate_sensitivity(n, 9, path_offset=off)
            e.set_launch(pairs_in_flight=1)
            assert e.simulate_sensitivity(n, 9, path_offset=off) == pytest.approx(v1, rel=1e-12)
            e.set_launch(pairs_in_flight=2)
        np.testing.assert_allclose(e.simulate_ZBC_control_variate(3000, 5, S1=4.97), o.zbc_moments(P, f, 3000, 5, S1=4.97), rtol=REL_F32)   # 497 steps: tail of 1
        runs = e.simulate_ZBC_runs(6000, 1234, n_runs=3)
        for k in range(3):
            assert np.array_equal(runs[k], e.simulate_ZBC_control_variate(6000, 1234 + k * 12345))
        # automatic: one pair below two chip-fills, two from there on -- same sums either way
        e.set_launch(pairs_in_flight=-1)
        cus = e.device_info()["compute_units"]
        n_big = 2 * 2048 * cus + 333
        assert e.launch_geometry("zbc", n_big)["grid"] * 2 * 512 >= n_big > e.launch_geometry("zbc", n_big)["grid"] * 512
        assert e.launch_geometry("zbc", n_big - 1000)["grid"] * 512 >= n_big - 1000
        auto = e.simulate_ZBC_control_variate(n_big, 3)
        e.set_launch(pairs_in_flight=1)
        np.testing.assert_allclose(auto, e.simulate_ZBC_control_variate(n_big, 3), rtol=1e-13)
    finally:
        e.set_launch(-1, 0, 0, -1)


def test_zbc_edge_cases_and_sharding(setup, hwmc_mod):
    e, o, P, f = setup("head")
    assert np.all(e.simulate_ZBC_control_variate(0, 1) == 0.0)                    # empty
    one = e.simulate_ZBC_control_variate(1, 1)                                    # single pair
    np.testing.assert_allclose(one, o.zbc_moments(P, f, 1, 1), rtol=REL_F32)
    # ragged sizes around the block / wave boundaries, huge 64-bit offsets
    for n in (63, 64, 65, 255, 257, 1000):
        got = e.simulate_ZBC_control_variate(n, 3, path_offset=2 ** 40 + 17)
        np.testing.assert_allclose(got, o.zbc_moments(P, f, n, 3, path_offset=2 ** 40 + 17), rtol=REL_F32)
    # waves that straddle a multiple of 2^32 pair ids (the lanes' high id words differ: the kernels then leave the
    # scalar round-2 multiply of philox_block_prio<UHI> for the per-lane form) and waves just beside one
    for off in (2 ** 32 - 30, 2 ** 32 - 64, 2 ** 32, 2 ** 33 - 1000):
        got = e.simulate_ZBC_control_variate(2000, 5, path_offset=off)
        np.testing.assert_allclose(got, o.zbc_moments(P, f, 2000, 5, path_offset=off), rtol=REL_F32)
    np.testing.assert_allclose(e.simulate_zcb(300, 5, path_offset=2 ** 32 - 100),
                               o.zcb_curve(300, 5, path_offset=2 ** 32 - 100), rtol=REL_F32)
    assert e.simulate_sensitivity(300, 5, path_offset=2 ** 32 - 100) == pytest.approx(
        o.vega_pathwise(P, f, 300, 5, path_offset=2 ** 32 - 100), rel=REL_F32)
    # shards by path_offset add up to the whole (what the multi-GPU path relies on)
    whole = e.simulate_ZBC_control_variate(10000, 9)
    parts = sum(e.simulate_ZBC_control_variate(m, 9, path_offset=off)
                for off, m in ((0, 2500), (2500, 2500), (5000, 1234), (6234, 3766)))
    np.testing.assert_allclose(whole, parts, rtol=1e-13)
    # bit-reproducible: the reduction order is fixed
    again = e.simulate_ZBC_control_variate(10000, 9)
    assert np.array_equal(whole, again)
    # a different seed gives different numbers
    assert not np.array_equal(whole, e.simulate_ZBC_control_variate(10000, 10))


def test_zbc_batched_runs_equal_separate_launches(setup, hwmc_mod):
    """The 20-run validation in one launch (hw_zbc_moments_runs): every run bit-identical to its own launch."""
    for prec in ("f32", "f64"):
        e, o, P, f = setup("head", prec)
        for n in (1000, 1 << 14):
            runs = e.simulate_ZBC_runs(n, 555, n_runs=20, seed_stride=12345, path_offset=77)
            assert runs.shape == (20, 5)
            for k in (0, 1, 7, 19):
                single = e.simulate_ZBC_control_variate(n, 555 + 12345 * k, path_offset=77)
                assert np.array_equal(runs[k], single)
        np.testing.assert_allclose(runs[3], o.zbc_moments(P, f, 1 << 14, 555 + 3 * 12345, path_offset=77),
                                   rtol=REL_F32 if prec == "f32" else REL_F64)
    e, _, P, f = setup("readme")
    big, ms = e.simulate_ZBC_runs(1 << 20, 1234, return_ms=True)                   # 20 x 2^21 paths, one launch
    prices = [hwmc_mod.cv_price(s, 2 << 20, float(P[-1]))["price"] for s in big]
    st = hwmc_mod.run_statistics(prices)
    assert abs(st["mean"] - 0.03544557) < 3 * st["se"] + 2e-6 and st["cv_percent"] < 0.15   # README: CV 0.0496 %
    with pytest.raises(hwmc_mod.HwError):
        e.simulate_ZBC_runs(16, 1, n_runs=0)


def test_zbc_other_option_and_grid(setup, hwmc_mod):
    """S1/S2/K other than the hard-coded 5/10/e^-0.1, a 4000-step grid (2000 executed steps) and a
    step count that is not a multiple of 4 (tail of the 4-normals-per-Philox-block loop)."""
    e, o, P, f = setup("head")
    got = e.simulate_ZBC_control_variate(4096, 11, S1=2.5, S2=7.3, K=0.8)
    np.testing.assert_allclose(got, o.zbc_moments(P, f, 4096, 11, S1=2.5, S2=7.3, K=0.8), rtol=REL_F32)
    got = e.simulate_ZBC_control_variate(4096, 11, S1=4.97, S2=9.0, K=0.9)        # 497 steps
    assert o.steps_to(4.97) == 497
    np.testing.assert_allclose(got, o.zbc_moments(P, f, 4096, 11, S1=4.97, S2=9.0, K=0.9), rtol=REL_F32)
    e4, o4, P4, f4 = setup("head", "f32", 4000)
    for st in (hwmc_mod.HW_STEPPER_SUPERPOSED, hwmc_mod.HW_STEPPER_REFERENCE):
        got = e4.simulate_ZBC_control_variate(2048, 13, stepper=st)
        np.testing.assert_allclose(got, o4.zbc_moments(P4, f4, 2048, 13), rtol=REL_F32)


def test_error_behaviour(hwmc_mod):
    p = hwmc_mod.default_params("head")
    e = hwmc_mod.HullWhiteEngine(p, device=0)
    try:
        with pytest.raises(hwmc_mod.HwError) as ei:                # "Did you run Q1 first?"
            e.simulate_ZBC_control_variate(16, 1)
        assert ei.value.code == -3
        P, f = hwmc_mod.analytic_curve(p)
        with pytest.raises(hwmc_mod.HwError):                      # wrong curve length
            e.load_market_data(P[:50], f[:50])
        e.load_market_data(P, f)
        for bad in (dict(S1=0.0), dict(S1=6.0, S2=5.0), dict(S2=11.0)):
            with pytest.raises(hwmc_mod.HwError):
                e.simulate_ZBC_control_variate(16, 1, **bad)
        with pytest.raises(hwmc_mod.HwError):
            e.simulate_ZBC_control_variate(-1, 1)
        with pytest.raises(hwmc_mod.HwError):
            e.set_launch(100, 0, 0, 1)
    finally:
        e.close()
    bad = hwmc_mod.default_params("head", n_steps=1001)            # common.cuh:25-27 static check
    with pytest.raises(hwmc_mod.HwError):
        hwmc_mod.HullWhiteEngine(bad, device=0)


# ---- curve kernel -------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_zcb_curve_vs_oracle(setup, hwmc_mod, golden, precision):
    e, o, P, f = setup("head", precision)
    tol = REL_F32 if precision == "f32" else REL_F64
    for n, seed in ((1 << 10, 1234), (3000, 5)):
        got = e.simulate_zcb(n, seed)
        want = o.zcb_curve(n, seed)
        assert got[0] == 2.0 * n                                   # path count: exact
        np.testing.assert_allclose(got, want, rtol=tol)
    for case in golden["cases"]:
        if "zcb_P_sum" in case and case["profile"] == "head" and case["precision"] == precision:
            got = e.simulate_zcb(case["n"], case["seed"])
            np.testing.assert_allclose(got, case["zcb_P_sum"], rtol=tol)
    # host post-processing of the curve equals the oracle's (compute_average_and_forward)
    c = e.simulate_zcb(1024, 1234)
    Pg, fg = e.compute_average_and_forward(c, 2048)
    Po, fo = o.average_and_forward(c, 2048)
    assert Pg[0] == 1.0
    np.testing.assert_array_equal(Pg, Po)
    np.testing.assert_allclose(fg, fo, rtol=0, atol=1e-7)
    # sharding + reproducibility
    whole = e.simulate_zcb(5000, 21)
    parts = e.simulate_zcb(2000, 21) + e.simulate_zcb(3000, 21, path_offset=2000)
    # float32 engine: the 64 lanes of a wave are summed in float32 before going to double, so a
    # different grouping of pairs into waves moves the sum by float rounding; float64: exact.
    np.testing.assert_allclose(whole[1:], parts[1:], rtol=1e-6 if precision == "f32" else 1e-13)
    assert np.array_equal(whole, e.simulate_zcb(5000, 21))


def test_zcb_curve_other_sigma_and_grid(setup):
    e, o, P, f = setup("head")
    got = e.simulate_zcb(2048, 3, sigma=0.101)                     # recompute_market_data, 3_sens:449-482
    np.testing.assert_allclose(got, o.zcb_curve(2048, 3, sigma=0.101), rtol=REL_F32)
    e4, o4, _, _ = setup("head", "f32", 4000)
    np.testing.assert_allclose(e4.simulate_zcb(512, 3), o4.zcb_curve(512, 3), rtol=REL_F32)


# ---- Vega ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_vega_pathwise_vs_oracle(setup, golden, precision):
    e, o, P, f = setup("head", precision)
    # the indicator 1{P>K} can flip on a path within rounding of the strike: a flip moves the sum
    # by ~0.3, i.e. ~1e-4 relative at 2^14 paths; none occurs for these seeds, and REL_F32 holds.
    tol = REL_F32 if precision == "f32" else REL_F64
    for n, seed in ((1 << 10, 1234), (1 << 14, 20240613), (777, 3)):
        got = e.simulate_sensitivity(n, seed)
        assert got == pytest.approx(o.vega_pathwise(P, f, n, seed), rel=tol)
    for case in golden["cases"]:
        if case["profile"] == "head" and case["precision"] == precision:
            got = e.simulate_sensitivity(case["n"], case["seed"], path_offset=case["path_offset"])
            assert got == pytest.approx(case["vega_sum"], rel=tol)
    a = e.simulate_sensitivity(4096, 5, antithetic=True)
    b = e.simulate_sensitivity(4096, 5)
    assert a != b and abs(a / 8192 - b / 4096) < 0.02


def test_vega_batched_runs_equal_separate_launches(setup):
    """The 20-run Vega bootstrap in one launch (hw_vega_pathwise_runs, 3_sens:527-654): every run bit-identical
    to its own launch, f32 and f64, ragged count and an offset."""
    for prec in ("f32", "f64"):
        e, o, P, f = setup("readme", prec)
        runs = e.simulate_sensitivity_runs(3001, 77, n_runs=20, seed_stride=982451653, path_offset=500)
        for k in (0, 1, 7, 19):
            assert runs[k] == e.simulate_sensitivity(3001, 77 + k * 982451653, path_offset=500)
        assert len(set(runs.tolist())) == 20
    anti = e.simulate_sensitivity_runs(1000, 5, n_runs=3, seed_stride=11, antithetic=True)
    assert anti[2] == e.simulate_sensitivity(1000, 5 + 22, antithetic=True)


def test_finite_difference_vega(setup, hwmc_mod, golden):
    e, o, P, f = setup("head")
    n, seed, eps = 1 << 14, 1234, 1e-3
    s0 = o.sigma
    want = []
    for sg in (s0 - eps, s0 + eps):
        sm = o.zbc_moments(P, f, n, seed, sigma=sg, drift=o.shifted_drift(sg, s0))
        want.append(sm)
    sc = e.simulate_ZBC_scenarios(n, seed, [s0 - eps, s0 + eps], [hwmc_mod.HW_DRIFT_SHIFTED] * 2)
    np.testing.assert_allclose(sc[0], want[0], rtol=REL_F32)
    np.testing.assert_allclose(sc[1], want[1], rtol=REL_F32)
    fused = e.run_finite_difference(n, seed, eps, fused=True)
    split = e.run_finite_difference(n, seed, eps, fused=False)
    assert fused["vega_fd"] == pytest.approx(split["vega_fd"], rel=1e-9)        # same draws, same arithmetic
    pw = [hwmc_mod.cv_price(w, 2 * n, float(P[-1]))["price"] for w in want]
    # a price difference of 5e-4 carries the 1e-5 relative price tolerance amplified ~50x
    assert fused["vega_fd"] == pytest.approx((pw[1] - pw[0]) / (2 * eps), rel=1e-3)
    for case in golden["cases"]:
        if "zbc_sums_fd" in case and case["profile"] == "head" and case["precision"] == "f32":
            sc = e.simulate_ZBC_scenarios(case["n"], case["seed"], [s0 - eps, s0, s0 + eps],
                                          [hwmc_mod.HW_DRIFT_SHIFTED, hwmc_mod.HW_DRIFT_BASE, hwmc_mod.HW_DRIFT_SHIFTED])
            np.testing.assert_allclose(sc[0], case["zbc_sums_fd"][0], rtol=REL_F32)
            np.testing.assert_allclose(sc[2], case["zbc_sums_fd"][1], rtol=REL_F32)
            np.testing.assert_allclose(sc[1], case["zbc_sums"], rtol=REL_F32)


def test_finite_difference_recalibrated_vs_oracle(setup, hwmc_mod):
    """run_finite_difference_recalibrated (3_sens:449-525): curves re-simulated at sigma -+ eps, then
    the option priced on them -- the same sequence on the oracle."""
    e, o, P, f = setup("head")
    n, seed, eps = 1 << 13, 11, 1e-3
    s0 = o.sigma
    got = e.run_finite_difference_recalibrated(n, seed, eps)
    want = []
    for sg in (s0 - eps, s0 + eps):
        Ps, fs = o.average_and_forward(o.zcb_curve(n, seed, sigma=sg), 2 * n)
        sums = o.zbc_moments(Ps, fs, n, seed, sigma=sg)
        want.append(o.cv_price(sums, 2 * n, float(Ps[-1]))["price"])
    # f(0,S1) is a float32 difference of logs of the re-simulated curve (market_data.cuh:125): a
    # 1e-7 relative difference in P_sum moves f by ~3e-5 relative and the price by ~1e-5 through A(t,T)
    assert got["p_minus"] == pytest.approx(want[0], rel=1e-4)
    assert got["p_plus"] == pytest.approx(want[1], rel=1e-4)
    assert e.P is P or np.array_equal(e.P, P)                      # market restored
    # the documented pathology: about twice the true Vega (README.md:51, SURVEY 6.2: 0.476 for 'head')
    big = e.run_finite_difference_recalibrated(1 << 18, 5)
    assert 0.3 < big["vega_fd_recalibrated"] < 0.7


def test_sample_paths_vs_oracle(setup, golden):
    e, o, _, _ = setup("head")
    got = e.simulate_paths_show(32, 4242)
    assert got.shape == (32, 1001) and got.dtype == np.float32      # analyze.py:15-27 layout
    want = o.sample_paths(32, 4242)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    g = golden["sample_paths_head"]
    np.testing.assert_allclose(got[:3][:, g["r_at"]], np.array(g["r"]), rtol=0, atol=2e-6)
    assert np.all(got[:, 0] == np.float32(0.012))


# ---- BASELINE.json sizes: size-independent properties ---------------------------------------------
def test_full_size_properties_2pow20(setup, hwmc_mod):
    """Q2b at 2^20 pairs x 1000-step grid (configs[1]) and Q3 at 2^20 paths (configs[2])."""
    e, o, P, f = setup("readme")
    n = 1 << 20
    whole, ms = e.simulate_ZBC_control_variate(n, 1234, return_ms=True)
    assert np.array_equal(whole, e.simulate_ZBC_control_variate(n, 1234))           # reproducible
    halves = sum(e.simulate_ZBC_control_variate(n // 2, 1234, path_offset=k * (n // 2)) for k in range(2))
    np.testing.assert_allclose(whole, halves, rtol=1e-13)                           # shards add up
    eighths = sum(e.simulate_ZBC_control_variate(n // 8, 1234, path_offset=k * (n // 8)) for k in range(8))
    np.testing.assert_allclose(whole, eighths, rtol=1e-13)
    ref = e.simulate_ZBC_control_variate(n, 1234, stepper=hwmc_mod.HW_STEPPER_REFERENCE)
    np.testing.assert_allclose(whole, ref, rtol=REL_F32)                            # two steppers agree
    # the oracle on the SAME 2^20 pairs (about a second of OpenMP work on the GPU box's host cores)
    np.testing.assert_allclose(whole, o.zbc_moments(P, f, n, 1234), rtol=REL_F32)
    ow = hwmc_mod.cv_price(o.zbc_moments(P, f, n, 1234), 2 * n, float(P[-1]))
    gw = hwmc_mod.cv_price(whole, 2 * n, float(P[-1]))
    assert gw["price"] == pytest.approx(ow["price"], rel=REL_F32) and gw["beta"] == pytest.approx(ow["beta"], rel=REL_F32)
    r = hwmc_mod.cv_price(whole, 2 * n, float(P[-1]))
    assert abs(r["price"] - 0.03544557) < 4 * r["ci95_halfwidth"] / 1.96            # exact Jamshidian
    assert r["beta"] == pytest.approx(0.166309, abs=1e-3)
    assert r["rho"] == pytest.approx(0.6738, abs=2e-3)
    vega = e.simulate_sensitivity(n, 99) / n
    assert vega == pytest.approx(0.230244, abs=1.5e-3)                              # analytic Vega
    assert vega == pytest.approx(o.vega_pathwise(P, f, n, 99) / n, rel=REL_F32)     # oracle, same 2^20 paths
    fd = e.run_finite_difference(n, 99)
    assert fd["vega_fd"] == pytest.approx(0.230244, abs=2e-3)
    assert abs(fd["vega_fd"] - vega) / vega < 0.02                                  # README.md:49-50: 0.18 %
    c = e.simulate_zcb(n, 5)
    assert c[0] == 2.0 * n
    Pmc, fmc = e.compute_average_and_forward(c, 2 * n)
    assert float(np.max(np.abs(Pmc - P))) < 4e-4                                    # ~4 SE at 2^21 paths
    assert ms > 0


# ---- in-process communicator (hw_comm_*: shard + one grouped ncclAllReduce) ---------------------
def test_comm_one_device_equals_single_context(setup, hwmc_mod):
    """Only one GPU is visible to this suite, so the communicator has one rank: the RCCL all-reduce is
    then an identity on the staged sums and every result must equal the single-context call BIT FOR BIT
    (the >1 rank arithmetic -- contiguous shards by global path id, sums of doubles -- is what
    test_zbc_edge_cases_and_sharding / test_full_size_properties_2pow20 check through path_offset)."""
    e, o, P, f = setup("readme")
    n = 50_001
    with hwmc_mod.HullWhiteFleet([e]) as fleet:
        a, ms = fleet.simulate_ZBC_control_variate(n, 77, return_ms=True)
        assert np.array_equal(a, e.simulate_ZBC_control_variate(n, 77)) and ms > 0
        np.testing.assert_allclose(a, o.zbc_moments(P, f, n, 77), rtol=REL_F32)
        assert np.array_equal(fleet.simulate_ZBC_control_variate(n, 77, path_offset=1000),
                              e.simulate_ZBC_control_variate(n, 77, path_offset=1000))
        c = fleet.simulate_zcb(n, 5)
        assert np.array_equal(c, e.simulate_zcb(n, 5)) and c[0] == 2.0 * n
        assert fleet.simulate_sensitivity(n, 9) == e.simulate_sensitivity(n, 9)
        assert np.array_equal(fleet.simulate_ZBC_runs(n, 1234, n_runs=20), e.simulate_ZBC_runs(n, 1234, n_runs=20))   # 100 doubles, one all-reduce
        assert np.array_equal(fleet.simulate_ZBC_control_variate(0, 1), np.zeros(5))
    with pytest.raises(hwmc_mod.HwError):
        hwmc_mod.HullWhiteFleet([e, e])                                             # one device twice


def test_comm_threaded_dispatch_rehearsal_with_four_workers(hwmc_mod, oracle_mod):
    """The per-device worker threads of hw_comm_* (one persistent host thread per context: spin, then sleep on a condition
    variable; the job runs on the worker, its error text travels back) with G = 4 -- on ONE GPU, which the
    HW_COMM_HOST_SUM mode allows: no RCCL, the four contexts share device 0, the library adds the four shards' sums in
    rank order.  Every call must equal the four shard calls added in that order BIT FOR BIT, and the single-context sums
    to 1e-13; 300 back-to-back calls (the spin path) and calls after a pause (the sleep path); errors raised on a worker
    reach the caller with their text."""
    import time
    p = hwmc_mod.default_params("readme")
    P, f = hwmc_mod.analytic_curve(p)
    engines = [hwmc_mod.HullWhiteEngine(p, device=0) for _ in range(4)]
    for e in engines:
        e.load_market_data(P, f)
    ref = engines[0]
    try:
        with hwmc_mod.HullWhiteFleet(engines, host_sum=True) as fleet:
            for n in (1001, 50_001, 1 << 18, 3):                                        # 3 pairs over 4 workers: one gets none
                got = fleet.simulate_ZBC_control_variate(n, 77)
                shards = [hwmc_mod.parallel.shard(n, g, 4) for g in range(4)]
                want = np.zeros(5)
                for off, cnt in shards:                                                # rank order, as the library adds them
                    want = want + ref.simulate_ZBC_control_variate(cnt, 77, path_offset=off)
                assert np.array_equal(got, want), n
                np.testing.assert_allclose(got, ref.simulate_ZBC_control_variate(n, 77), rtol=1e-13)
            c = fleet.simulate_zcb(20_000, 5)
            assert c[0] == 2.0 * 20_000
            np.testing.assert_allclose(c[1:], ref.simulate_zcb(20_000, 5)[1:], rtol=1e-6)   # float row sums regroup with the shards
            assert fleet.simulate_sensitivity(40_000, 9) == pytest.approx(ref.simulate_sensitivity(40_000, 9), rel=1e-13)
            runs = fleet.simulate_ZBC_runs(30_001, 1234, n_runs=20, seed_stride=12345)   # the 20-run validation over the fleet
            np.testing.assert_allclose(runs, ref.simulate_ZBC_runs(30_001, 1234, n_runs=20, seed_stride=12345), rtol=1e-13)
            np.testing.assert_allclose(runs[7], ref.simulate_ZBC_control_variate(30_001, 1234 + 7 * 12345), rtol=1e-13)
            first = fleet.simulate_ZBC_control_variate(4096, 1)
            for i in range(300):                                                       # back to back: workers never sleep
                assert np.array_equal(fleet.simulate_ZBC_control_variate(4096, 1), first)
            for pause in (0.002, 0.05, 0.3):                                           # longer than the 100 us spin: futex path
                time.sleep(pause)
                assert np.array_equal(fleet.simulate_ZBC_control_variate(4096, 1), first)
            with pytest.raises(hwmc_mod.HwError) as ei:                                # S2 > T_final: refused on the calling thread,
                fleet.simulate_ZBC_control_variate(4096, 1, S1=5.0, S2=11.0)           # before any worker could sit in an all-reduce
            assert "S1 < S2" in str(ei.value) and "device" not in str(ei.value)
            with pytest.raises(hwmc_mod.HwError) as ei:                                # found by the workers only (launch geometry):
                fleet.simulate_ZBC_runs(1 << 27, 1, n_runs=200, seed_stride=1)         # nothing is launched, the text travels back
            assert "device 0" in str(ei.value) and "2^22" in str(ei.value)
            assert np.array_equal(fleet.simulate_ZBC_control_variate(4096, 1), first)   # and the communicator still works
        with pytest.raises(hwmc_mod.HwError):
            hwmc_mod.HullWhiteFleet([engines[0], engines[0]], host_sum=True)           # the same context twice
    finally:
        for e in engines:
            e.close()


# ---- curve kernel variants ---------------------------------------------------------------------------
def test_zcb_curve_blocked_kernel_variants(hwmc_mod, oracle_mod, monkeypatch):
    """The float curve runs a blocked kernel (half Philox blocks, P pairs per thread: P = 1 below 2^21 pairs, 2
    from 2^21, 4 from 2^22 on a 256-CU chip) and falls back to the per-step kernel when a maturity cannot fall on a
    half-block boundary (odd stride).  All of them against the oracle, and against the per-step kernel forced by
    HWMC_CURVE_GENERIC=1 (read when the context is created)."""
    p = hwmc_mod.default_params("readme")
    o = oracle_mod.Oracle("readme")
    e = hwmc_mod.HullWhiteEngine(p, device=0)
    monkeypatch.setenv("HWMC_CURVE_GENERIC", "1")
    g = hwmc_mod.HullWhiteEngine(p, device=0)
    monkeypatch.delenv("HWMC_CURVE_GENERIC")
    cus = e.device_info()["compute_units"]
    for n in (1000, 4 * 2048 * cus + 77, 8 * 2048 * cus + 5):                      # P = 1, 1, 2 by the engine's rule (ragged counts)
        got = e.simulate_zcb(n, 11)
        assert got[0] == 2.0 * n
        np.testing.assert_allclose(got, g.simulate_zcb(n, 11), rtol=2e-6)           # two float evaluation orders
        for forced in ("1", "2", "4"):                                             # ... and every P forced at every size class
            monkeypatch.setenv("HWMC_CURVE_P", forced)
            np.testing.assert_allclose(e.simulate_zcb(n, 11), got, rtol=2e-6, err_msg="P = %s, n = %d" % (forced, n))
        monkeypatch.delenv("HWMC_CURVE_P")
        assert e.launch_geometry("zcb_blocked", n)["units_per_thread"] == (2 if n >= 8 * 2048 * cus else 1)
        if n <= 4 * 2048 * cus + 77:
            np.testing.assert_allclose(got, o.zcb_curve(n, 11), rtol=REL_F32)
        half = n // 2                                                              # shards add up (other P per shard)
        parts = e.simulate_zcb(half, 11) + e.simulate_zcb(n - half, 11, path_offset=half)
        np.testing.assert_allclose(got[1:], parts[1:], rtol=1e-6)
    e.close()
    g.close()
    # odd stride (1000 steps, 201 maturities -> every 5 steps): per-step kernel
    p5 = hwmc_mod.default_params("readme", n_mat=201)
    e5 = hwmc_mod.HullWhiteEngine(p5, device=0)
    o5 = oracle_mod.Oracle("readme", n_mat=201)
    np.testing.assert_allclose(e5.simulate_zcb(2000, 4), o5.zcb_curve(2000, 4), rtol=REL_F32)
    e5.close()


@pytest.mark.parametrize("n_mat,n_steps", [(2, 1000), (3, 8), (11, 1000), (16, 60), (17, 64), (18, 68), (32, 124), (33, 1024),
                                           (51, 1000), (101, 4000), (126, 1000), (251, 1000)])
def test_zcb_curve_maturity_counts(hwmc_mod, oracle_mod, monkeypatch, n_mat, n_steps):
    """The blocked float kernel reduces maturities sixteen at a time (MaturityTree): group boundaries at 16, 32, ...,
    a partial last group completed with zero vectors, a single maturity, a group that ends exactly on the last one.
    Every maturity count against the oracle (1e-5) and against the per-step kernel with its per-maturity slots; P = 1, 2
    and 4 pairs per thread (HWMC_CURVE_P) on a ragged pair count; two passes through one workgroup (grid cap)."""
    p = hwmc_mod.default_params("readme", n_steps=n_steps, n_mat=n_mat)
    o = oracle_mod.Oracle("readme", n_steps=n_steps, n_mat=n_mat)
    n = 1500 + n_mat                                                   # ragged: not a multiple of 64
    want = o.zcb_curve(n, 17)
    assert want.shape == (n_mat,) and want[0] == 2.0 * n
    for P in ("1", "2", "4"):
        monkeypatch.setenv("HWMC_CURVE_P", P)
        with hwmc_mod.HullWhiteEngine(p, device=0) as e:
            got = e.simulate_zcb(n, 17)
            np.testing.assert_allclose(got, want, rtol=REL_F32, err_msg="P=%s" % P)
            e.set_launch(block_size=64, max_blocks_per_cu=0)           # one wave per workgroup
            np.testing.assert_allclose(e.simulate_zcb(n, 17), want, rtol=REL_F32)
            # a capped grid: every thread walks several paths, the tree is re-armed between them
            e.set_launch(block_size=128, max_blocks_per_cu=1)
            cap = e.device_info()["compute_units"] * 128
            big = 2 * cap * int(P) + 333
            got_big = e.simulate_zcb(big, 5)
            assert got_big[0] == 2.0 * big
            halves = e.simulate_zcb(cap * int(P), 5) + e.simulate_zcb(big - cap * int(P), 5, path_offset=cap * int(P))
            np.testing.assert_allclose(got_big[1:], halves[1:], rtol=1e-6)
    monkeypatch.delenv("HWMC_CURVE_P")
    monkeypatch.setenv("HWMC_CURVE_GENERIC", "1")
    with hwmc_mod.HullWhiteEngine(p, device=0) as g:
        np.testing.assert_allclose(g.simulate_zcb(n, 17), want, rtol=REL_F32)
    # the double engine: the per-step kernel (HWMC_CURVE_GENERIC still set) and the blocked kernel with its state in double
    pd = hwmc_mod.default_params("readme", n_steps=n_steps, n_mat=n_mat, precision="f64")
    od = oracle_mod.Oracle("readme", n_steps=n_steps, n_mat=n_mat, precision="f64")
    want64 = od.zcb_curve(n, 17)
    with hwmc_mod.HullWhiteEngine(pd, device=0) as d:
        np.testing.assert_allclose(d.simulate_zcb(n, 17), want64, rtol=REL_F64)
    monkeypatch.delenv("HWMC_CURVE_GENERIC")
    for P in ("1", "2", "4"):
        monkeypatch.setenv("HWMC_CURVE_P", P)
        with hwmc_mod.HullWhiteEngine(pd, device=0) as d:
            np.testing.assert_allclose(d.simulate_zcb(n, 17), want64, rtol=REL_F64, err_msg="f64 P=%s" % P)
            d.set_launch(block_size=64, max_blocks_per_cu=0)
            np.testing.assert_allclose(d.simulate_zcb(n, 17), want64, rtol=REL_F64)
            d.set_launch(block_size=128, max_blocks_per_cu=1)
            cap = d.device_info()["compute_units"] * 128
            big = 2 * cap * int(P) + 333
            got_big = d.simulate_zcb(big, 5)
            assert got_big[0] == 2.0 * big
            halves = d.simulate_zcb(cap * int(P), 5) + d.simulate_zcb(big - cap * int(P), 5, path_offset=cap * int(P))
            np.testing.assert_allclose(got_big[1:], halves[1:], rtol=1e-6)
    monkeypatch.delenv("HWMC_CURVE_P")


def test_zcb_curve_blocked_equals_generic_at_full_size(hwmc_mod, oracle_mod, monkeypatch):
    """The blocked float curve kernel (half Philox blocks, transposed reduction, P pairs per thread) against the per-step
    kernel with its own reduction -- two independent evaluation orders of the same float arithmetic -- at every size class
    up to BASELINE's 2^20 pairs and beyond (P = 1, 2, 4; 256- and 512-thread workgroups; ragged counts), and against
    the oracle itself at 2^20 pairs (about a second of OpenMP work)."""
    p = hwmc_mod.default_params("readme")
    monkeypatch.setenv("HWMC_CURVE_GENERIC", "1")
    g = hwmc_mod.HullWhiteEngine(p, device=0)
    monkeypatch.delenv("HWMC_CURVE_GENERIC")
    e = hwmc_mod.HullWhiteEngine(p, device=0)
    for n in (1000, 1 << 17, (1 << 18) + 512, 1 << 19, 1 << 20, (1 << 20) + 77, 1 << 21, 1 << 22):
        a, b = e.simulate_zcb(n, 99), g.simulate_zcb(n, 99)
        assert a[0] == b[0] == 2.0 * n
        np.testing.assert_allclose(a[1:], b[1:], rtol=2e-7, err_msg="n = %d" % n)
    o = oracle_mod.Oracle("readme")
    for sg in (0.099, 0.101):                                          # the two curves of the recalibrated FD (3_sens:449-525)
        np.testing.assert_allclose(e.simulate_zcb(1 << 20, 99, sigma=sg), o.zcb_curve(1 << 20, 99, sigma=sg), rtol=REL_F32)
    e.close()
    g.close()


def test_zcb_curve_f64_blocked_equals_generic_at_full_size(hwmc_mod, oracle_mod, monkeypatch):
    """The double engine's curve: the blocked kernel (state y, W in double, half Philox blocks, contributions through the
    float v_exp_f32, transposed reduction) against the per-step double kernel (double exp2, per-maturity slots in double)
    at every size class (P = 1, 2, 4), against the double oracle at 2^16 pairs, against the float engine at 2^20 pairs
    (same draws: float and double state differ by float rounding only), and under weak mean reversion, where the FLOAT
    engine must leave the (y, W) form but the double one need not."""
    p = hwmc_mod.default_params("readme", precision="f64")
    monkeypatch.setenv("HWMC_CURVE_GENERIC", "1")
    g = hwmc_mod.HullWhiteEngine(p, device=0)
    monkeypatch.delenv("HWMC_CURVE_GENERIC")
    e = hwmc_mod.HullWhiteEngine(p, device=0)
    assert e.kernel_info("zcb_blocked")["scratch_bytes"] == 0
    for n in (1000, (1 << 18) + 512, (1 << 20) + 77, 1 << 21, 1 << 22):
        a, b = e.simulate_zcb(n, 99), g.simulate_zcb(n, 99)
        assert a[0] == b[0] == 2.0 * n
        np.testing.assert_allclose(a[1:], b[1:], rtol=2e-7, err_msg="n = %d" % n)
    o = oracle_mod.Oracle("readme", precision="f64")
    np.testing.assert_allclose(e.simulate_zcb(1 << 16, 5), o.zcb_curve(1 << 16, 5), rtol=REL_F64)
    with hwmc_mod.HullWhiteEngine(hwmc_mod.default_params("readme"), device=0) as e32:
        np.testing.assert_allclose(e32.simulate_zcb(1 << 20, 99), e.simulate_zcb(1 << 20, 99), rtol=REL_F32)
    e.close()
    g.close()
    pw = hwmc_mod.default_params("readme", precision="f64")
    pw.a = 0.05                                                                    # (1 - e) sqrt(1000) = 0.016 < 0.05
    ow = oracle_mod.Oracle("readme", precision="f64")
    ow.m.a = 0.05
    with hwmc_mod.HullWhiteEngine(pw, device=0) as ew:
        np.testing.assert_allclose(ew.simulate_zcb(3000, 8), ow.zcb_curve(3000, 8), rtol=REL_F64)


def test_full_size_properties_of_the_multi_gpu_shares(setup, hwmc_mod):
    """One GPU's share of BASELINE configs[3] (2^24 pairs over 8 GPUs -> 2^21 per GPU) and configs[4] (2^26 pairs x
    4000-step grid over 8 GPUs -> 2^23 per GPU, f32 vs f64): the oracle cannot follow at these sizes in seconds, so
    the checks are size-independent properties -- shards add up to the whole, the first and the last rank's shard
    price the same option within Monte Carlo error of the closed form, f32 and f64 steppers agree on the same draws."""
    e, o, P, f = setup("readme")
    n = 1 << 21
    exact = 0.03544557                                                             # Jamshidian, readme profile
    first = e.simulate_ZBC_control_variate(n, 4321, path_offset=0)
    last = e.simulate_ZBC_control_variate(n, 4321, path_offset=7 * n)              # rank 7 of 8
    halves = sum(e.simulate_ZBC_control_variate(n // 2, 4321, path_offset=7 * n + k * (n // 2)) for k in range(2))
    np.testing.assert_allclose(last, halves, rtol=1e-13)
    assert not np.array_equal(first, last)
    for sums in (first, last, first + last):
        cnt = 2 * n * (2 if sums is not first and sums is not last else 1)
        r = hwmc_mod.cv_price(sums, cnt, float(P[-1]))
        assert abs(r["price"] - exact) < 4 * r["ci95_halfwidth"] / 1.96
    # stress share: 2^23 pairs on the 4000-step grid, both precisions, same seed -> same normals
    n5 = 1 << 23
    e32, _, P4, _ = setup("readme", "f32", 4000)
    e64, _, _, _ = setup("readme", "f64", 4000)
    s32 = e32.simulate_ZBC_control_variate(n5, 99, path_offset=3 * n5)
    s64 = e64.simulate_ZBC_control_variate(n5, 99, path_offset=3 * n5)
    np.testing.assert_allclose(s32, s64, rtol=REL_F32)
    r32, r64 = (hwmc_mod.cv_price(s, 2 * n5, float(P4[-1])) for s in (s32, s64))
    assert r32["price"] == pytest.approx(r64["price"], rel=REL_F32) and r32["beta"] == pytest.approx(r64["beta"], rel=REL_F32)
    assert abs(r64["price"] - exact) < 4 * r64["ci95_halfwidth"] / 1.96 + 2e-6     # + trapezoid bias at dt = 0.0025
    v32 = e32.simulate_sensitivity(n5, 7) / n5
    v64 = e64.simulate_sensitivity(n5, 7) / n5
    assert v32 == pytest.approx(v64, rel=REL_F32) and v64 == pytest.approx(0.230244, abs=1e-3)


def _random_models(n_models, seed=20250607):
    """The 40 seeded random models of the parity envelope: mean reversion, volatility, r0, both theta pieces and the break, grid
    lengths that are and are not multiples of four, maturity counts, exercise / maturity dates, strike factor around the forward,
    path count and seed (drawn in this order: the sequence is part of the committed evidence, profiles/r04_parity_envelope.txt)."""
    rng = np.random.default_rng(seed)
    for case in range(n_models):
        n_mat = int(rng.choice([11, 21, 51, 101]))
        per = int(rng.choice([2, 3, 4, 5, 8, 10, 12]))                     # steps per maturity: odd ones take the generic curve kernel
        n_steps = per * (n_mat - 1)
        T_final = float(rng.choice([5.0, 10.0, 20.0]))
        fields = dict(a=float(rng.uniform(0.3, 2.0)), sigma=float(rng.uniform(0.02, 0.2)), r0=float(rng.uniform(0.0, 0.05)),
                      T_final=T_final, th_break=float(rng.uniform(0.2, 0.8)) * T_final, th_a1=float(rng.uniform(0.0, 0.03)),
                      th_b1=float(rng.uniform(-0.001, 0.003)), th_a2=float(rng.uniform(0.0, 0.03)), th_b2=float(rng.uniform(-0.001, 0.003)))
        spacing = T_final / (n_mat - 1)
        i1 = int(rng.integers(1, n_mat - 1))
        S1, S2 = i1 * spacing, float(rng.uniform(i1 * spacing + 0.3 * spacing, T_final))
        k_factor = float(rng.uniform(0.97, 1.03))                          # strike around the money (times the forward bond price)
        n, sd = int(rng.integers(500, 6000)), int(rng.integers(1, 1 << 40))
        yield dict(case=case, n_mat=n_mat, n_steps=n_steps, T_final=T_final, fields=fields, spacing=spacing, S1=S1, S2=S2, k_factor=k_factor, n=n, seed=sd)


def test_random_models_against_the_double_oracle(hwmc_mod, oracle_mod):
    """The reference hard-codes one model (common.cuh:16-39); the C ABI takes it as parameters.  40 random models -- mean
    reversion, volatility, r0, both theta pieces and the break, grid lengths that are and are not multiples of four, maturity
    counts, exercise / maturity / strike -- through all three kernels, the FD scenarios, the batched runs and (float) the reference stepper:

      * DOUBLE engine against the double oracle at 2e-6 (same normals, so this is sharp: a wrong index, table entry or hoisted
        constant shows as 1e-3 or worse);
      * float engine against the same double sums at 5e-5;
      * float engine, BOTH steppers, against the FLOAT oracle -- the reference's own per-step float32 arithmetic -- at the
        north-star tolerance REL_F32 = 1e-5 wherever float32 arithmetic determines the sums that well.  It does not everywhere:
        at low volatility the payoff max(P - K, 0) is a difference of nearly equal numbers, and the float oracle ITSELF sits up
        to 3e-5 from the double oracle (c below; measured envelope, tools/labs/parity_envelope.py, profiles/r04_parity_envelope.txt:
        c > 1e-5 on 5 of the 40 models, all with sigma < 0.07).  So every model is held to max(1e-5, 2 c) -- the engine is as close
        to the reference's float arithmetic as that arithmetic is to the exact sums -- and at least 32 of the 40 must meet the plain
        1e-5 with both steppers (measured: 35; superposed stepper alone 36; worst 2.7e-5 / 3.3e-5 on the model with c = 3.0e-5);
        the curve meets the plain 1e-5 on all 40 (measured worst 3.4e-7);
      * pathwise Vega: a sum of terms of both signs, so its error is held against the scale of the terms,
        tol x (sum |term1| + sum |term2|) (3_sens:72-80; the oracle reports the two absolute sums): 2e-6 for the double engine
        (measured worst 5.3e-7), 1e-5 for the float engine against either oracle (measured worst 2.7e-6)."""
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b))))
    plain, worst = 0, {"superposed": (0.0, None), "reference": (0.0, None), "curve": (0.0, None), "vega_f32": (0.0, None), "vega_f64": (0.0, None)}

    def note(key, val, case):
        if val > worst[key][0]:
            worst[key] = (val, case)

    n_models = 40
    for m in _random_models(n_models):
        case, n_mat, n_steps, fields, spacing, S1, S2 = m["case"], m["n_mat"], m["n_steps"], m["fields"], m["spacing"], m["S1"], m["S2"]
        engines = {}
        try:
            for prec in ("f64", "f32"):
                p = hwmc_mod.default_params("head", n_steps=n_steps, n_mat=n_mat, precision=prec)
                for k, v in fields.items():
                    setattr(p, k, v)
                engines[prec] = hwmc_mod.HullWhiteEngine(p, device=0)
            P, f = hwmc_mod.analytic_curve(engines["f64"].params)
            o = oracle_mod.Oracle("head", n_steps=n_steps, n_mat=n_mat, precision="f64")
            o32 = oracle_mod.Oracle("head", n_steps=n_steps, n_mat=n_mat, precision="f32")
            for k, v in fields.items():
                setattr(o.m, k, v)
                setattr(o32.m, k, v)
            fwd = float(np.interp(S2, np.arange(n_mat) * spacing, P) / np.interp(S1, np.arange(n_mat) * spacing, P))
            K = fwd * m["k_factor"]                                         # around the money
            n, seed = m["n"], m["seed"]
            want = o.zbc_moments(P, f, n, seed, S1=S1, S2=S2, K=K)
            wantc = o.zcb_curve(n, seed)
            wantv, abs1, abs2 = o.vega_pathwise_terms(P, f, n, seed, S1=S1, S2=S2, K=K)
            vscale = abs1 + abs2                                            # the scale of the Vega sum's signed terms
            for prec, tol in (("f64", REL_F64), ("f32", 5e-5)):
                e = engines[prec]
                e.load_market_data(P, f)
                tag = (case, prec, fields, n_steps, n_mat, S1, S2, K)
                np.testing.assert_allclose(e.simulate_ZBC_control_variate(n, seed, S1=S1, S2=S2, K=K), want, rtol=tol, atol=tol * abs(want[1]) * 1e-3, err_msg=str(tag))
                np.testing.assert_allclose(e.simulate_zcb(n, seed), wantc, rtol=tol, err_msg=str(tag))
                got = e.simulate_sensitivity(n, seed, S1=S1, S2=S2, K=K)
                vtol = REL_F64 if prec == "f64" else REL_F32
                assert abs(got - wantv) <= vtol * vscale, (tag, got, wantv, vscale, abs(got - wantv) / vscale)
                note("vega_" + prec, abs(got - wantv) / vscale, case)
                # the finite-difference scenarios (sigma -+ eps with the shifted drift table, 3_sens:374-446) from ONE pass
                s0, eps = fields["sigma"], 1e-3
                sc = e.simulate_ZBC_scenarios(n, seed, [s0 - eps, s0 + eps], [hwmc_mod.HW_DRIFT_SHIFTED] * 2, S1=S1, S2=S2, K=K)
                for k, sg in enumerate((s0 - eps, s0 + eps)):
                    np.testing.assert_allclose(sc[k], o.zbc_moments(P, f, n, seed, S1=S1, S2=S2, K=K, sigma=sg, drift=o.shifted_drift(sg, s0)),
                                               rtol=tol, atol=tol * abs(want[1]) * 1e-3, err_msg=str(tag))
                np.testing.assert_allclose(e.simulate_ZBC_runs(n, seed, n_runs=3, seed_stride=977, S1=S1, S2=S2, K=K)[2],
                                           o.zbc_moments(P, f, n, seed + 2 * 977, S1=S1, S2=S2, K=K), rtol=tol, atol=tol * abs(want[1]) * 1e-3, err_msg=str(tag))
            # the reference stepper (float only): evolve_hull_white_step verbatim with the drift table on the device
            e32 = engines["f32"]
            sup = e32.simulate_ZBC_control_variate(n, seed, S1=S1, S2=S2, K=K)
            ref = e32.simulate_ZBC_control_variate(n, seed, S1=S1, S2=S2, K=K, stepper=hwmc_mod.HW_STEPPER_REFERENCE)
            np.testing.assert_allclose(ref, want, rtol=5e-5, atol=5e-8 * abs(want[1]), err_msg=str((case, "reference stepper", fields)))
            # ---- the float engine against the FLOAT oracle (the reference's arithmetic) at the north-star tolerance
            want32 = o32.zbc_moments(P, f, n, seed, S1=S1, S2=S2, K=K)
            c = rel(want32, want)                                           # how well float32 arithmetic determines these sums at all
            d_sup, d_ref = rel(sup, want32), rel(ref, want32)
            bound = max(REL_F32, 2.0 * c)
            assert d_sup <= bound and d_ref <= bound, (case, fields, "superposed %.2e reference %.2e vs float oracle; float oracle vs double oracle %.2e" % (d_sup, d_ref, c))
            plain += d_sup <= REL_F32 and d_ref <= REL_F32
            note("superposed", d_sup, case)
            note("reference", d_ref, case)
            d_curve = rel(e32.simulate_zcb(n, seed)[1:], o32.zcb_curve(n, seed)[1:])
            assert d_curve <= REL_F32, (case, fields, d_curve)
            note("curve", d_curve, case)
            v32, a1, a2 = o32.vega_pathwise_terms(P, f, n, seed, S1=S1, S2=S2, K=K)
            got32 = e32.simulate_sensitivity(n, seed, S1=S1, S2=S2, K=K)
            assert abs(got32 - v32) <= REL_F32 * (a1 + a2), (case, fields, got32, v32, a1 + a2)
        finally:
            for e in engines.values():
                e.close()
    assert plain >= 32, "only %d of %d models meet the plain 1e-5 against the float oracle with both steppers; worst cases %r" % (plain, n_models, worst)
    print("parity envelope over %d random models: %d at the plain 1e-5 (both steppers); worst (value, case): %r" % (n_models, plain, worst))


def test_random_models_at_scale_against_the_reference_arithmetic(hwmc_mod):
    """The same 40 random models where the per-path rounding scatter no longer matters: 2^24 pairs each, the production engine
    (superposed recursion, tables from double) against the reference's OWN float arithmetic on the same draws
    (HW_STEPPER_REFERENCE = evolve_hull_white_step verbatim, common.cuh:237-244, on the reference's float drift table,
    common.cuh:60-84 -- both pinned against the CPU oracle at small n by the test above).  At a few thousand paths five
    low-volatility models sit 1-3e-5 apart (the envelope test above); averaged over 3.4e7 paths the north star's quantities --
    price and beta* -- agree to 1e-5 on ALL 40 models (measured worst: price 7.5e-6, on a deep out-of-the-money option worth
    3e-4; sums 9.9e-6), and the double engine says which side is closer to the truth: the reference's arithmetic is up to 1.8e-5
    from it, the engine up to 1.3e-5, on the same model (1200 steps, sigma 0.05, price 1e-3)."""
    n = 1 << 24
    relmax = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b))))
    worst = {"price": (0.0, None), "beta": (0.0, None), "sums": (0.0, None), "engine_vs_double": (0.0, None), "reference_vs_double": (0.0, None)}
    for m in _random_models(40):
        grid_T = np.arange(m["n_mat"]) * m["spacing"]
        res = {}
        for name, prec, tables, stepper in (("double", "f64", "f64", hwmc_mod.HW_STEPPER_SUPERPOSED), ("engine", "f32", "f64", hwmc_mod.HW_STEPPER_SUPERPOSED),
                                            ("reference", "f32", "ref_f32", hwmc_mod.HW_STEPPER_REFERENCE)):
            p = hwmc_mod.default_params("head", n_steps=m["n_steps"], n_mat=m["n_mat"], precision=prec, table_mode=tables)
            for k, v in m["fields"].items():
                setattr(p, k, v)
            with hwmc_mod.HullWhiteEngine(p, device=0) as e:
                if name == "double":
                    P, f = hwmc_mod.analytic_curve(p)
                    P_S2 = float(np.interp(m["S2"], grid_T, P))
                    K = P_S2 / float(np.interp(m["S1"], grid_T, P)) * m["k_factor"]
                e.load_market_data(P, f)
                sums = e.simulate_ZBC_control_variate(n, m["seed"], S1=m["S1"], S2=m["S2"], K=K, stepper=stepper)
            res[name] = (sums, hwmc_mod.cv_price(sums, 2 * n, P_S2))
        pd = abs(res["double"][1]["price"])
        vals = {"price": abs(res["engine"][1]["price"] - res["reference"][1]["price"]) / pd,
                "beta": abs(res["engine"][1]["beta"] - res["reference"][1]["beta"]) / abs(res["double"][1]["beta"]),
                "sums": relmax(res["engine"][0], res["reference"][0]),
                "engine_vs_double": relmax(res["engine"][0], res["double"][0]), "reference_vs_double": relmax(res["reference"][0], res["double"][0])}
        for k, v in vals.items():
            if v > worst[k][0]:
                worst[k] = (v, m["case"])
        ctx = (m["case"], m["fields"], m["n_steps"], vals)
        assert vals["price"] <= REL_F32 and vals["beta"] <= REL_F32, ctx                  # the north star's quantities at its tolerance
        assert vals["sums"] <= 1.5e-5 and vals["engine_vs_double"] <= 3e-5 and vals["reference_vs_double"] <= 3e-5, ctx
    print("40 random models x 2^24 pairs, engine vs the reference's float arithmetic; worst (value, case): %r" % worst)


def test_configs_3_and_4_whole_on_one_gpu(setup, hwmc_mod):
    """BASELINE configs[3] (Q2b, 2^24 pairs, 8 GPUs, all-reduce of the moments) and configs[4] (2^26 pairs x 4000-step grid,
    f32 vs f64, 8 GPUs) at their FULL path counts -- on the one GPU this suite has: the sample set depends only on
    (seed, global pair id), so what eight devices would compute is, shard by shard, what eight calls with the shards'
    offsets compute here, and the all-reduce is a sum of eight 5-vectors.  Checked: the eight shard calls add up to the
    whole-batch call (1e-13); the in-process fleet with EIGHT worker threads (HW_COMM_HOST_SUM: the contexts share the
    device, the library adds the shards in rank order) returns exactly that sum, bit for bit; the price sits within 4 SE of
    the closed form at 2^25 / 2^27 paths (SE 3.5e-6 / 1.7e-6: a bias of 1e-5 would show); float and double steppers agree on
    the same draws at the stress size.  No multi-GPU box has been available: this is the arithmetic of those two
    configurations without the xGMI hop."""
    exact = 0.03544557                                                             # Jamshidian, readme profile
    e, _, P, f = setup("readme")
    n, G = 1 << 24, 8
    whole = e.simulate_ZBC_control_variate(n, 20240613)
    shards = [e.simulate_ZBC_control_variate(n // G, 20240613, path_offset=g * (n // G)) for g in range(G)]
    in_rank_order = np.zeros(5)
    for sh in shards:
        in_rank_order = in_rank_order + sh
    np.testing.assert_allclose(whole, in_rank_order, rtol=1e-13)
    p = hwmc_mod.default_params("readme")
    engines = [hwmc_mod.HullWhiteEngine(p, device=0) for _ in range(G)]
    try:
        for x in engines:
            x.load_market_data(P, f)
        with hwmc_mod.HullWhiteFleet(engines, host_sum=True) as fleet:
            got, ms = fleet.simulate_ZBC_control_variate(n, 20240613, return_ms=True)
            assert np.array_equal(got, in_rank_order) and ms > 0
            runs = fleet.simulate_ZBC_runs(1 << 21, 1234, n_runs=20, seed_stride=12345)        # the 20-run validation, sharded
            np.testing.assert_allclose(runs, e.simulate_ZBC_runs(1 << 21, 1234, n_runs=20, seed_stride=12345), rtol=1e-13)
    finally:
        for x in engines:
            x.close()
    r = hwmc_mod.cv_price(whole, 2 * n, float(P[-1]))
    assert abs(r["price"] - exact) < 4 * r["ci95_halfwidth"] / 1.96, (r["price"], r["ci95_halfwidth"])
    assert r["beta"] == pytest.approx(0.166309, abs=3e-4) and r["rho"] == pytest.approx(0.6738, abs=5e-4)
    # configs[4]: the stress size, both precisions
    n5 = 1 << 26
    e32, _, P4, _ = setup("readme", "f32", 4000)
    e64, _, _, _ = setup("readme", "f64", 4000)
    s32 = e32.simulate_ZBC_control_variate(n5, 99)
    s64 = e64.simulate_ZBC_control_variate(n5, 99)
    np.testing.assert_allclose(s32, s64, rtol=REL_F32)
    eighth = sum(e32.simulate_ZBC_control_variate(n5 // G, 99, path_offset=g * (n5 // G)) for g in range(G))
    np.testing.assert_allclose(s32, eighth, rtol=1e-13)
    r32, r64 = (hwmc_mod.cv_price(x, 2 * n5, float(P4[-1])) for x in (s32, s64))
    assert r32["price"] == pytest.approx(r64["price"], rel=REL_F32) and r32["beta"] == pytest.approx(r64["beta"], rel=REL_F32)
    assert abs(r64["price"] - exact) < 4 * r64["ci95_halfwidth"] / 1.96 + 2e-6, (r64["price"], r64["ci95_halfwidth"])   # + trapezoid bias at dt = 0.0025


def test_curve_kernel_at_the_stress_size(setup, hwmc_mod):
    """BASELINE configs[4] for the Q1 kernel: simulate_zcb (include/market_data.cuh:25-79) at 2^26 pairs on the 4000-step grid,
    float and double engine -- the launch bench.py times as stress.{f32,f64}.curve (8 192 workgroups x 8 passes x two pairs
    per thread), which no test ran above 2^22 pairs (missing 3).  No oracle finishes this size, so the properties
    that do not depend on it: the path count is exact (P_sum[0] = 2n); the eight shards of an 8-GPU run add up to the whole
    (1e-13 double, 1e-9 float: both kernels add rows of 16 lanes in float before they go to double, and a shard of 2^23 pairs
    is launched as exactly one of the whole launch's eight passes -- same grid, same lanes together --, so only the order of
    the double additions differs; the float bar would also hold if a later geometry grouped the lanes differently); float and double
    agree to 1e-5 on every maturity (same draws); every P(0,T) sits within 4 SE of the exact curve (SE 6e-6 at T = 10: a bias
    of 3e-5 would show); the launch really is the many-pass, two-pairs-per-thread shape."""
    n, G = 1 << 26, 8
    e32, _, P4, _ = setup("readme", "f32", 4000)
    e64, _, _, _ = setup("readme", "f64", 4000)
    p = e32.params
    res = {}
    for name, e, shard_tol in (("f32", e32, 1e-9), ("f64", e64, 1e-13)):
        g = e.launch_geometry("zcb_blocked", n)
        assert g["iters"] > 1 and g["units_per_thread"] == 2 and g["grid"] * g["block"] * g["iters"] * 2 >= n, g
        whole = e.simulate_zcb(n, 1234)
        assert whole.shape == (p.n_mat,) and whole[0] == 2.0 * n                    # integer path count, exact
        parts = np.zeros(p.n_mat)
        for k in range(G):
            parts = parts + e.simulate_zcb(n // G, 1234, path_offset=k * (n // G))
        assert parts[0] == 2.0 * n
        np.testing.assert_allclose(whole[1:], parts[1:], rtol=shard_tol, err_msg=name)
        assert np.array_equal(whole, e.simulate_zcb(n, 1234))                       # fixed reduction order: bit-reproducible
        res[name] = whole
    np.testing.assert_allclose(res["f32"][1:], res["f64"][1:], rtol=REL_F32)
    se = hwmc_mod.curve_standard_errors(p, P4, n)
    for name in ("f32", "f64"):
        Pmc = res[name] / (2.0 * n)
        z = np.abs(Pmc[1:] - np.asarray(P4, np.float64)[1:]) / (se[1:] + 2e-7)      # + float32 storage of the exact curve
        assert z.max() < 4.0, (name, int(z.argmax()) + 1, float(z.max()), float(Pmc[-1]), float(P4[-1]))
        Pq, fq = e32.compute_average_and_forward(res[name], 2 * n)
        assert Pq[0] == 1.0 and float(Pq[-1]) == pytest.approx(float(P4[-1]), abs=4 * se[-1] + 2e-7)


def test_more_than_2_to_the_31_pairs_in_one_call(setup, hwmc_mod):
    """Maximum sizes: the C ABI counts pairs in int64 and the kernels index them in 64 bits -- one call with 2^31 + 77 pairs
    (4.3e9 paths: 32 times BASELINE's largest configuration; the grid is capped at 2^20 workgroups and the threads loop) must
    (i) count its paths exactly -- the curve kernel's P_sum[0] = 2n = 2^32 + 154, not a value wrapped at 32 bits --, (ii) equal
    the sum of three shard calls that meet at 2^30 and 2^31 (1e-13: same sample set, another order of the double additions),
    (iii) price the option within 4 SE of the closed form with NO allowance for what a float path and the trapezoid rule may
    be off by: at 4.3e9 paths SE = 5.5e-7 (1.6e-5 of the price), so a systematic error of the float engine beyond ~2e-6 would
    show (measured: 1.3e-7 from the closed form, 0.23 SE).  No per-path state exists, so nothing grows with n except time:
    0.35 s per ZBC call, 0.8 s for the curve."""
    e, _, P, f = setup("readme")
    n = (1 << 31) + 77
    cuts = (0, 1 << 30, 1 << 31, n)
    whole = e.simulate_ZBC_control_variate(n, 20240613)
    parts = np.zeros(5)
    for a, b in zip(cuts, cuts[1:]):
        parts = parts + e.simulate_ZBC_control_variate(b - a, 20240613, path_offset=a)
    np.testing.assert_allclose(whole, parts, rtol=1e-13)
    r = hwmc_mod.cv_price(whole, 2 * n, float(P[-1]))
    se = r["ci95_halfwidth"] / 1.96
    assert se < 1e-6 and abs(r["price"] - 0.03544557) < 4 * se, (r["price"], se)
    assert r["beta"] == pytest.approx(0.166309, abs=1e-4) and r["rho"] == pytest.approx(0.6738, abs=5e-4)   # (measured 0.166324, 0.67340)
    g = e.launch_geometry("zbc", n)
    assert g["grid"] <= 1 << 20 and g["iters"] >= 2 and g["grid"] * g["block"] * g["iters"] * g["units_per_thread"] >= n
    curve = e.simulate_zcb(n, 7)
    assert curve[0] == 2.0 * n == float((1 << 32) + 154)                               # exact, beyond 32 bits
    assert abs(curve[-1] / (2.0 * n) - float(P[-1])) < 4 * hwmc_mod.curve_standard_errors(e.params, P, n)[-1] + 2e-6
    v = e.simulate_sensitivity(n, 11)
    v_parts = sum(e.simulate_sensitivity(b - a, 11, path_offset=a) for a, b in zip(cuts, cuts[1:]))
    assert v == pytest.approx(v_parts, rel=1e-12) and v / n == pytest.approx(0.230244, abs=2e-4)
    print("2^31 + 77 pairs in one call: price %.9f (closed form 0.035445570; SE %.2e), beta* %.6f, rho %.5f, P(0,10) %.8f (exact %.8f), "
          "pathwise Vega %.6f (closed form 0.230244)" % (r["price"], se, r["beta"], r["rho"], curve[-1] / (2.0 * n), float(P[-1]), v / n))


@pytest.mark.parametrize("profile", ["readme", "head"])
def test_systematic_distance_to_the_reference_arithmetic_at_scale(hwmc_mod, profile):
    """North star: 'prices ... must match the reference CUDA run on the same seeds within 1e-5 relative'.  No reference run
    exists to compare with, but its ARITHMETIC does: HW_STEPPER_REFERENCE is evolve_hull_white_step verbatim
    (common.cuh:237-244) on the reference's float drift table (HW_TABLES_REF_F32, common.cuh:60-84), bit-checked against the
    oracle elsewhere in this file.  On the same 2^31 paths (2^30 pairs: the per-path rounding noise, ~1e-7 with random sign, has
    averaged out) the price of the production engine -- superposed recursion, tables from double -- must sit within 2e-6 of the
    price of that arithmetic, i.e. a fifth of the north star's tolerance, and both within 2e-6 of the double engine's; measured
    at 2^33 pairs 4.3e-7 / 9.4e-7 (tools/bias_probe.py, profiles/r05_bias_probe.txt)."""
    n, seed = 1 << 30, 20240613
    price = {}
    for name, prec, tables, stepper in (("engine", "f32", "f64", hwmc_mod.HW_STEPPER_SUPERPOSED), ("reference_arithmetic", "f32", "ref_f32", hwmc_mod.HW_STEPPER_REFERENCE),
                                        ("double", "f64", "f64", hwmc_mod.HW_STEPPER_SUPERPOSED)):
        p = hwmc_mod.default_params(profile, precision=prec, table_mode=tables)
        P, f = hwmc_mod.analytic_curve(p)
        with hwmc_mod.HullWhiteEngine(p, device=0) as e:
            e.load_market_data(P, f)
            price[name] = hwmc_mod.cv_price(e.simulate_ZBC_control_variate(n, seed, stepper=stepper), 2 * n, float(P[-1]))["price"]
    rel = lambda a, b: abs(price[a] - price[b]) / price["double"]
    print("%s, 2^30 pairs: engine - reference arithmetic %.1e, engine - double %.1e, reference arithmetic - double %.1e of the price"
          % (profile, rel("engine", "reference_arithmetic"), rel("engine", "double"), rel("reference_arithmetic", "double")))
    assert rel("engine", "reference_arithmetic") < 2e-6 and rel("engine", "double") < 2e-6 and rel("reference_arithmetic", "double") < 2e-6, price
    # the other two kernels, float against double engine on the same draws: pathwise Vega on 2^30 paths (a sum of signed terms:
    # measured 9e-7 / 4e-7 of the Vega), the curve on 2^28 pairs (every maturity; measured 4e-8)
    vega, curve = {}, {}
    for prec in ("f32", "f64"):
        p = hwmc_mod.default_params(profile, precision=prec)
        P, f = hwmc_mod.analytic_curve(p)
        with hwmc_mod.HullWhiteEngine(p, device=0) as e:
            e.load_market_data(P, f)
            vega[prec] = e.simulate_sensitivity(n, 99) / n
            curve[prec] = e.simulate_zcb(1 << 28, 5)
    assert abs(vega["f32"] - vega["f64"]) < 3e-6 * vega["f64"], vega
    assert vega["f64"] == pytest.approx(hwmc_mod.closed_form_vega(p, P), abs=8e-5)          # 4 SE at 2^30 paths (sd of one path's integrand 0.58)
    np.testing.assert_allclose(curve["f32"][1:], curve["f64"][1:], rtol=2e-7)
    assert curve["f32"][0] == curve["f64"][0] == 2.0 * (1 << 28)


def test_weak_mean_reversion_falls_back_to_the_per_step_form(hwmc_mod, oracle_mod):
    """The float (y, W) form of the time loop divides by 1 - e and is used only while (1 - e) sqrt(steps) >= 0.05
    (ou_walk).  Away from the reference's a = 1 every float32 implementation -- the float32 oracle included -- sits
    ~1e-5 from the double oracle, so the bar here is 5e-5 against the DOUBLE oracle on the same draws: a = 0.02 and
    0.1 run the per-step fallback, a = 0.5 the blocked form; the reference stepper must agree as well."""
    n = 20000
    for a in (0.02, 0.1, 0.5):
        p = hwmc_mod.default_params("readme")
        p.a = a
        o64 = oracle_mod.Oracle("readme", precision="f64")
        o64.m.a = a
        e = hwmc_mod.HullWhiteEngine(p, device=0)
        P, f = hwmc_mod.analytic_curve(p)
        e.load_market_data(P, f)
        want = o64.zbc_moments(P, f, n, 3)
        np.testing.assert_allclose(e.simulate_ZBC_control_variate(n, 3), want, rtol=5e-5)
        np.testing.assert_allclose(e.simulate_ZBC_control_variate(n, 3, stepper=hwmc_mod.HW_STEPPER_REFERENCE), want, rtol=5e-5)
        assert e.simulate_sensitivity(n, 4) == pytest.approx(o64.vega_pathwise(P, f, n, 4), rel=2e-4)
        np.testing.assert_allclose(e.simulate_zcb(n, 5), o64.zcb_curve(n, 5), rtol=5e-5)
        e.close()


def test_random_parameter_sweep_against_the_double_oracle(hwmc_mod, oracle_mod):
    """24 seeded random parameter sets (mean reversion, volatility, grid, maturity grid, option dates, strike): the
    float32 engine -- whichever kernel variant the parameters select (blocked or per-step time loop, blocked or generic
    curve kernel, tails of 1..3 steps, P pairs per thread) -- stays within 1e-4 of the DOUBLE oracle on the same
    draws and the float64 engine within 2e-6, for the ZBC moments, the Vega sum and the curve."""
    rng = np.random.default_rng(20240613)
    for case in range(24):
        n_mat = int(rng.choice([11, 21, 51, 101]))
        n_steps = (n_mat - 1) * int(rng.choice([2, 5, 8, 10, 25, 40]))
        a = float(rng.choice([0.05, 0.3, 1.0, 2.5]))
        sigma = float(rng.uniform(0.02, 0.2))
        S1 = max(float(rng.uniform(0.3, 7.0)), 2.0 * 10.0 / n_steps)                # at least two time steps
        S2 = float(rng.uniform(S1 + 0.5, 10.0))
        seed = int(rng.integers(1, 2 ** 40))
        n = int(rng.integers(500, 3000))
        res = {}
        for prec in ("f32", "f64"):
            p = hwmc_mod.default_params("readme", n_steps=n_steps, n_mat=n_mat, precision=prec, sigma=sigma)
            p.a = a
            e = hwmc_mod.HullWhiteEngine(p, device=0)
            P, f = hwmc_mod.analytic_curve(p)
            e.load_market_data(P, f)
            o = oracle_mod.Oracle("readme", n_steps=n_steps, n_mat=n_mat, sigma=sigma, precision="f64")
            o.m.a = a
            K = float(np.interp(S2, np.linspace(0, 10, n_mat), P) / np.interp(S1, np.linspace(0, 10, n_mat), P))   # at the money forward
            tol = 1e-4 if prec == "f32" else 2e-6
            ctx = "case %d %s: n_steps=%d n_mat=%d a=%.2f sigma=%.3f S1=%.3f S2=%.3f n=%d" % (case, prec, n_steps, n_mat, a, sigma, S1, S2, n)
            got = e.simulate_ZBC_control_variate(n, seed, S1=S1, S2=S2, K=K)
            assert np.all(np.isfinite(got)), ctx
            np.testing.assert_allclose(got, o.zbc_moments(P, f, n, seed, S1=S1, S2=S2, K=K), rtol=tol, err_msg=ctx)
            v = e.simulate_sensitivity(n, seed + 1, S1=S1, S2=S2, K=K)
            vo, abs1, abs2 = o.vega_pathwise_terms(P, f, n, seed + 1, S1=S1, S2=S2, K=K)
            assert abs(v - vo) <= tol * (abs1 + abs2), (ctx, abs(v - vo) / (abs1 + abs2))   # a sum of signed terms: against the scale of the terms
            np.testing.assert_allclose(e.simulate_zcb(n, seed + 2), o.zcb_curve(n, seed + 2), rtol=tol, err_msg=ctx)
            e.close()


def test_random_seeds_offsets_and_counts(setup, hwmc_mod):
    """40 seeded random (seed, path_offset, count) triples at the reference configuration -- 64-bit seeds, offsets
    anywhere in 2^64 including just below multiples of 2^32, ragged counts -- ZBC moments of both steppers, the
    two-scenario FD launch and the batched runs against the float32 oracle at the north-star tolerance."""
    e, o, P, f = setup("readme")
    rng = np.random.default_rng(7)
    for case in range(40):
        seed = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        kind = case % 4
        if kind == 0:
            off = int(rng.integers(0, 2 ** 63)) * 2
        elif kind == 1:
            off = int(rng.integers(1, 2 ** 31)) * 2 ** 32 - int(rng.integers(1, 200))     # a wave straddles 2^32 k
        elif kind == 2:
            off = int(rng.integers(0, 2 ** 20))
        else:
            off = 2 ** 64 - 1 - int(rng.integers(5000, 10 ** 6))                          # near the top of the id space
        n = int(rng.integers(1, 700))
        ctx = "case %d: seed=%d off=%d n=%d" % (case, seed, off, n)
        want = o.zbc_moments(P, f, n, seed, path_offset=off)
        np.testing.assert_allclose(e.simulate_ZBC_control_variate(n, seed, path_offset=off), want, rtol=REL_F32, err_msg=ctx)
        if case % 5 == 0:
            np.testing.assert_allclose(e.simulate_ZBC_control_variate(n, seed, path_offset=off, stepper=hwmc_mod.HW_STEPPER_REFERENCE),
                                       want, rtol=REL_F32, err_msg=ctx)
            runs = e.simulate_ZBC_runs(n, seed % 2 ** 62, n_runs=3, seed_stride=12345, path_offset=off)
            np.testing.assert_allclose(runs[1], o.zbc_moments(P, f, n, seed % 2 ** 62 + 12345, path_offset=off), rtol=REL_F32, err_msg=ctx)


def test_single_launch_ticket_finish(setup, hwmc_mod, monkeypatch):
    """HWMC_SINGLE_LAUNCH=1 (read by hw_create): the grid-wide sums are finished inside the path kernel by the last
    workgroup to arrive (sharded tickets, self re-arming).  Same sums as the default two-launch finish to rounding,
    bit-reproducible, for grids from one workgroup to thousands, many launches in a row."""
    e, o, P, f = setup("readme")
    monkeypatch.setenv("HWMC_SINGLE_LAUNCH", "1")
    p = hwmc_mod.default_params("readme")
    t = hwmc_mod.HullWhiteEngine(p, device=0)
    monkeypatch.delenv("HWMC_SINGLE_LAUNCH")
    t.load_market_data(P, f)
    for n in (1, 200, 256 * 7, 256 * 8 + 3, 100_000, (1 << 20) + 77):            # 1, 1, 7, 9, 391, 1025 workgroups (the last: two pairs per thread)
        a = t.simulate_ZBC_control_variate(n, 31)
        np.testing.assert_allclose(a, e.simulate_ZBC_control_variate(n, 31), rtol=1e-13)
        assert np.array_equal(a, t.simulate_ZBC_control_variate(n, 31))             # tickets re-armed, fixed order
    for k in range(12):                                                            # back to back, varying sizes
        n = 1000 + 977 * k
        np.testing.assert_allclose(t.simulate_ZBC_control_variate(n, k), e.simulate_ZBC_control_variate(n, k), rtol=1e-13)
    assert t.simulate_sensitivity(50_000, 9) == pytest.approx(e.simulate_sensitivity(50_000, 9), rel=1e-13)
    sig = [0.099, 0.101]
    dm = [hwmc_mod.HW_DRIFT_SHIFTED] * 2
    np.testing.assert_allclose(t.simulate_ZBC_scenarios(30_000, 5, sig, dm), e.simulate_ZBC_scenarios(30_000, 5, sig, dm), rtol=1e-13)
    np.testing.assert_allclose(t.simulate_ZBC_control_variate(30_000, 5, stepper=hwmc_mod.HW_STEPPER_REFERENCE),
                               e.simulate_ZBC_control_variate(30_000, 5, stepper=hwmc_mod.HW_STEPPER_REFERENCE), rtol=1e-13)
    t.close()
