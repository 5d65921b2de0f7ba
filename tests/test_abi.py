"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/hw_mc.h declares, and its host-only entry points behave (no GPU compute here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hw_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol(hwmc_mod):
    """Both headers against the library and the Python binding: include/hw_mc.h is the stable boundary (what
    INTEGRATION.md binds), include/hw_mc_diag.h the knobs / probes / profiling hooks; together they are EXACTLY what
    libhw_hip.so exports under hw_*, and no diagnostic name sits in the stable header."""
    assert os.path.exists(hwmc_mod.LIB_PATH)
    stable, diag = _declared_symbols("hw_mc.h"), _declared_symbols("hw_mc_diag.h")
    assert len(stable) >= 24 and not set(stable) & set(diag)
    assert sorted(hwmc_mod.SIGNATURES) == stable          # python binding covers the whole stable header ...
    assert sorted(hwmc_mod.DIAG_SIGNATURES) == diag       # ... and the whole diagnostic one
    assert not [n for n in stable if n.startswith(("hw_debug_", "hw_profile_")) or n in ("hw_set_launch", "hw_get_kernel_info", "hw_get_launch_geometry")]
    lib = C.CDLL(hwmc_mod.LIB_PATH)
    for name in stable + diag:
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", hwmc_mod.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\bT (hw_[a-z0-9_]+)\b", out)))
    assert exported == sorted(stable + diag)              # nothing exported that no header declares
    version = int(re.search(r"#define HW_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "hw_mc.h")).read()).group(1))
    assert lib.hw_abi_version() == version == hwmc_mod._lib_abi_version()


def test_library_contains_gfx950_code_object(hwmc_mod):
    data = open(hwmc_mod.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"zbc_superposed_kernel" in data


def test_host_only_entry_points(hwmc_mod):
    p = hwmc_mod.default_params("head")
    assert (p.a, p.sigma, p.r0, p.T_final, p.n_steps, p.n_mat) == (1.0, 0.1, 0.012, 10.0, 1000, 101)
    assert p.th_a2 == 0.019 and hwmc_mod.default_params("readme").th_a2 == 0.014
    with pytest.raises(hwmc_mod.HwError):
        hwmc_mod.default_params("nope")
    P, f = hwmc_mod.analytic_curve(p)
    assert P[0] == 1.0 and float(P[50]) == pytest.approx(0.947126, abs=1e-6)
    assert float(f[0]) == pytest.approx(0.012, abs=1e-7)
    r = hwmc_mod.cv_price([2.0, 4.0, 3.0, 9.0, 5.0], 2, 1.9)
    assert r["mean_X"] == 1.0 and r["mean_Y"] == 2.0 and r["beta"] == pytest.approx(1.0)
    assert r["price"] == pytest.approx(1.0 - 0.1)


def test_host_post_processing_matches_oracle(hwmc_mod, oracle_mod):
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("head")
    c = o.zcb_curve(512, 5)
    Po, fo = o.average_and_forward(c, 1024)
    Pe = np.zeros(101, np.float32)
    fe = np.zeros(101, np.float32)
    lib = C.CDLL(hwmc_mod.LIB_PATH)
    dp = C.POINTER(C.c_double)
    fp = C.POINTER(C.c_float)
    rc = lib.hw_average_and_forward(C.byref(p), c.ctypes.data_as(dp), C.c_int64(1024),
                                    Pe.ctypes.data_as(fp), fe.ctypes.data_as(fp))
    assert rc == 0
    np.testing.assert_array_equal(Pe, Po)
    np.testing.assert_allclose(fe, fo, rtol=0, atol=1e-7)
    s = o.zbc_moments(P, f, 4096, 9)
    ro = o.cv_price(s, 8192, float(P[-1]))
    re_ = hwmc_mod.cv_price(s, 8192, float(P[-1]))
    for k in ("price", "beta", "rho", "mean_X", "mean_Y"):
        assert re_[k] == pytest.approx(ro[k], rel=2e-5)       # float32 vs double post-processing


@pytest.mark.parametrize("profile", ["head", "readme"])
def test_recover_theta_equals_the_oracle_elementwise(hwmc_mod, oracle_mod, profile):
    """hw_recover_theta (recover_theta + compute_derivative + theta_func: src/2_option_pricing.cu:14-35,
    include/common.cuh:228-230, 250-258) beside the oracle's restatement, element by element, on the exact curve
    and on a Monte Carlo curve (the oracle's own simulate_zcb at 2048 pairs).  Both are float32 host code calling
    glibc's expf with contraction off, so the three arrays must be IDENTICAL, not merely close."""
    p = hwmc_mod.default_params(profile)
    o = oracle_mod.Oracle(profile)
    _, f_exact = hwmc_mod.analytic_curve(p)
    _, f_mc = o.average_and_forward(o.zcb_curve(2048, 11), 4096)
    lib = C.CDLL(hwmc_mod.LIB_PATH)
    fp = C.POINTER(C.c_float)
    for f in (f_exact, f_mc):
        rec, orig, Ts = (np.zeros(101, np.float32) for _ in range(3))
        rc = lib.hw_recover_theta(C.byref(p), f.ctypes.data_as(fp), rec.ctypes.data_as(fp), orig.ctypes.data_as(fp),
                                  Ts.ctypes.data_as(fp))
        assert rc == 0
        rec_o, orig_o, Ts_o = o.recover_theta(f)
        np.testing.assert_array_equal(Ts, Ts_o)
        np.testing.assert_array_equal(orig, orig_o)
        np.testing.assert_array_equal(rec, rec_o)
    # and the statistic the reference prints (2_option:49-67): max |rec - orig| over every 10th point
    rec, orig, _ = o.recover_theta(f_exact)
    assert float(np.abs(rec - orig)[::10].max()) < 0.01


def test_no_gpu_gives_error_not_fallback(hwmc_mod):
    """Without a GPU hw_create must fail loudly (HW_E_HIP); with one, it must succeed."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hwmc_mod.HwError) as ei:
        hwmc_mod.HullWhiteEngine(hwmc_mod.default_params("head"))
    assert ei.value.code == -2
