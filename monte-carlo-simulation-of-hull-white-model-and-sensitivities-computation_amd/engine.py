"""Host-side mirror of the reference's hot-path interface, on top of the C ABI (include/hw_mc.h).

Method names follow the reference's functions so that tests read like the reference's drivers:

    compute_constants()              include/common.cuh:92-110        -> HullWhiteEngine(...)
    load_market_data_to_device()     include/common.cuh:171-176       -> .load_market_data(P, f)
    simulate_ZBC_control_variate     include/common.cuh:286-409       -> .simulate_ZBC_control_variate(...)
    simulate_zcb                     include/market_data.cuh:25-79    -> .simulate_zcb(...)
    compute_average_and_forward      include/market_data.cuh:101-127  -> .compute_average_and_forward(...)
    simulate_sensitivity             src/3_sensitivity_analysis.cu:22 -> .simulate_sensitivity(...)
    run_finite_difference            src/3_sensitivity_analysis.cu:400-> .run_finite_difference(...)
    simulate_paths_show              include/market_data.cuh:136-160  -> .simulate_paths_show(...)
    recover_theta                    src/2_option_pricing.cu:14-35    -> .recover_theta(f)

All compute goes through libhw_hip.so; nothing here falls back to NumPy.
"""
import ctypes as C
import math

import numpy as np

from . import _lib as L

K_DEFAULT = float(np.exp(np.float32(-0.1)))   # expf(-0.1f), src/2_option_pricing.cu:110


def default_params(profile="head", n_steps=1000, n_mat=101, precision="f32", table_mode="f64",
                   sigma=None):
    p = L.HwParams()
    L.check(L.load().hw_default_params(C.byref(p), profile.encode()))
    p.n_steps = n_steps
    p.n_mat = n_mat
    p.precision = {"f32": L.HW_F32, "f64": L.HW_F64}[precision]
    p.table_mode = {"f64": L.HW_TABLES_F64, "ref_f32": L.HW_TABLES_REF_F32}[table_mode]
    if sigma is not None:
        p.sigma = sigma
    return p


def analytic_curve(params):
    """Exact P(0,T), f(0,T) on the params' maturity grid (float32, like data/P.bin, data/f.bin)."""
    P = np.zeros(params.n_mat, np.float32)
    f = np.zeros(params.n_mat, np.float32)
    L.check(L.load().hw_analytic_curve(C.byref(params), _fp(P), _fp(f)))
    return P, f


def cv_price(sums, n_total, P0S2):
    """beta*, control-variate adjusted price (src/2_option_pricing.cu:154-175, 277-281)."""
    sums = np.ascontiguousarray(sums, np.float64)
    res = np.zeros(9, np.float64)
    L.check(L.load().hw_cv_price(_dp(sums), int(n_total), float(P0S2), _dp(res)))
    keys = ("mean_X", "mean_Y", "var_Y", "cov", "beta", "adjustment", "price", "var_X", "rho")
    out = dict(zip(keys, (float(v) for v in res)))
    # single-run normal-theory half-width of the CV estimator (additive)
    resid_var = max(out["var_X"] - out["cov"] ** 2 / out["var_Y"], 0.0)
    out["ci95_halfwidth"] = 1.96 * math.sqrt(resid_var / n_total)
    return out


def closed_form_zbc(params, P, S1=5.0, S2=10.0, K=None, sigma=None):
    """Exact price of the call on P(S1, S2) with strike K under Hull-White (Jamshidian), from the market curve P on the
    params' maturity grid (S1, S2 on grid points) -- what simulate_ZBC_control_variate (include/common.cuh:286-409) estimates.
    Its derivative in sigma with the curve held fixed is what both Vega estimators of src/3_sensitivity_analysis.cu target."""
    a = params.a
    sig = params.sigma if sigma is None else sigma
    K = K_DEFAULT if K is None else K
    step = params.T_final / (params.n_mat - 1)
    i1, i2 = int(round(S1 / step)), int(round(S2 / step))
    if abs(i1 * step - S1) > 1e-9 or abs(i2 * step - S2) > 1e-9:
        raise ValueError("closed_form_zbc: S1, S2 must be maturity grid points")
    P1, P2 = float(P[i1]), float(P[i2])
    B = (1.0 - math.exp(-a * (S2 - S1))) / a
    sp = sig * math.sqrt((1.0 - math.exp(-2.0 * a * S1)) / (2.0 * a)) * B
    h = math.log(P2 / (K * P1)) / sp + 0.5 * sp
    cdf = lambda x: 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))
    return P2 * cdf(h) - K * P1 * cdf(h - sp)


def closed_form_vega(params, P, S1=5.0, S2=10.0, K=None, h=1e-5):
    s0 = params.sigma
    return (closed_form_zbc(params, P, S1, S2, K, s0 + h) - closed_form_zbc(params, P, S1, S2, K, s0 - h)) / (2.0 * h)


def curve_standard_errors(params, P, n_pairs):
    """Standard error of the antithetic-pair Monte Carlo estimate of P(0, T_j) (simulate_zcb, include/market_data.cuh:25-79):
    D = exp(-I) is lognormal with Var I = V(T) = sigma^2/a^2 (T - 2B + (1 - e^{-2aT})/(2a)); the pair mean (D+ + D-)/2 has
    variance P^2 (cosh V - 1)."""
    a, s = params.a, params.sigma
    T = np.linspace(0.0, params.T_final, params.n_mat)
    B = (1.0 - np.exp(-a * T)) / a
    V = s * s / (a * a) * (T - 2.0 * B + (1.0 - np.exp(-2.0 * a * T)) / (2.0 * a))
    return np.asarray(P, np.float64) * np.sqrt(np.maximum(np.cosh(V) - 1.0, 0.0) / n_pairs)


def run_statistics(samples):
    """20-run statistics of src/2_option_pricing.cu:305-324 (t = 2.093, df = 19)."""
    x = np.asarray(samples, np.float64)
    n = len(x)
    mean = float(x.mean())
    var = float(((x - mean) ** 2).sum() / (n - 1))
    sd = math.sqrt(var)
    se = sd / math.sqrt(n)
    margin = 2.093 * se
    return {"mean": mean, "variance": var, "sd": sd, "se": se, "margin": margin,
            "ci_lo": mean - margin, "ci_hi": mean + margin, "cv_percent": 100.0 * sd / mean,
            "ci95_width": 2.0 * margin}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class HullWhiteEngine:
    """One GPU context (one device, one stream)."""

    def __init__(self, params=None, device=0, **kw):
        self.lib = L.load()
        self.params = params if params is not None else default_params(**kw)
        self._ctx = C.c_void_p()
        L.check(self.lib.hw_create(C.byref(self.params), int(device), C.byref(self._ctx)))
        self.P = self.f = None

    # -- lifecycle -------------------------------------------------------------------------
    def close(self):
        if self._ctx:
            self.lib.hw_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self):
        d = L.HwDeviceInfo()
        L.check(self.lib.hw_get_device_info(self._ctx, C.byref(d)))
        return {"name": d.name.decode(), "device": d.device, "compute_units": d.compute_units,
                "clock_mhz": d.clock_mhz, "wavefront_size": d.wavefront_size,
                "lds_per_cu_bytes": d.lds_per_cu_bytes, "hbm_bytes": d.hbm_bytes}

    def kernel_info(self, kernel="zbc"):
        k = L.HwKernelInfo()
        L.check(self.lib.hw_get_kernel_info(self._ctx, kernel.encode(), C.byref(k)))
        return {f: getattr(k, f) for f, _ in L.HwKernelInfo._fields_}

    def profile_enable(self, max_launches):
        """Record a hipEvent pair around the path kernel of the next `max_launches` launches."""
        L.check(self.lib.hw_profile_enable(self._ctx, int(max_launches)))

    def profile_read(self, cap=100000):
        ms = np.zeros(cap, np.float32)
        n = C.c_int32(0)
        L.check(self.lib.hw_profile_read(self._ctx, _fp(ms), cap, C.byref(n)))
        return ms[:n.value].copy()

    def profile_spans(self, ref=None, cap=100000):
        """(start_ms, end_ms) of every profiled launch, relative to the first profiled launch of `ref` (default: self).
        Call before profile_read()."""
        st = np.zeros(cap, np.float32)
        en = np.zeros(cap, np.float32)
        n = C.c_int32(0)
        L.check(self.lib.hw_profile_spans(self._ctx, (ref or self)._ctx, _fp(st), _fp(en), cap, C.byref(n)))
        return st[:n.value].copy(), en[:n.value].copy()

    def launch_geometry(self, kernel, n_units):
        b, g, it, pb = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int64(0)
        L.check(self.lib.hw_get_launch_geometry(self._ctx, kernel.encode(), int(n_units), C.byref(b), C.byref(g),
                                                C.byref(it), C.byref(pb)))
        slots = b.value * g.value * it.value
        return {"block": b.value, "grid": g.value, "iters": it.value, "partial_bytes": pb.value,
                "units_per_thread": max(1, -(-int(n_units) // slots)) if slots else 1}   # 2: two pairs per thread (effective_ilp)

    def profile_clock_mhz(self):
        """Shader clock held during the last profiled superposed-ZBC launch (in-kernel probe)."""
        mhz = C.c_double(0.0)
        L.check(self.lib.hw_profile_clock(self._ctx, C.byref(mhz)))
        return mhz.value

    def profile_clock_samples(self):
        """(mhz, start_us, end_us) of every probing workgroup of the last profiled ZBC / blocked-curve launch."""
        mhz, t0, t1 = (np.zeros(64, np.float64) for _ in range(3))
        n = C.c_int32(0)
        L.check(self.lib.hw_profile_clock_samples(self._ctx, _dp(mhz), _dp(t0), _dp(t1), 64, C.byref(n)))
        return mhz[:n.value].copy(), t0[:n.value].copy(), t1[:n.value].copy()

    def set_launch(self, block_size=0, max_blocks_per_cu=0, table_in_lds=0, pairs_in_flight=0):
        L.check(self.lib.hw_set_launch(self._ctx, block_size, max_blocks_per_cu, table_in_lds,
                                       pairs_in_flight))

    # -- inputs ----------------------------------------------------------------------------
    def load_market_data(self, P, f):
        self.P = np.ascontiguousarray(P, np.float32)
        self.f = np.ascontiguousarray(f, np.float32)
        L.check(self.lib.hw_set_market(self._ctx, _fp(self.P), _fp(self.f), len(self.P)))

    def drift_tables(self, sigma=0.0, drift_mode=L.HW_DRIFT_BASE):
        n = self.params.n_steps
        d = np.zeros(n, np.float64)
        sd = np.zeros(n, np.float64)
        L.check(self.lib.hw_drift_tables(self._ctx, float(sigma), drift_mode, _dp(d), _dp(sd)))
        return d, sd

    def _request(self, n_pairs, seed, path_offset, S1, S2, K, sigma, drift_mode, stepper):
        r = L.HwZbcRequest()
        r.n_pairs, r.seed, r.path_offset = int(n_pairs), int(seed), int(path_offset)
        r.S1, r.S2, r.K = float(S1), float(S2), float(K_DEFAULT if K is None else K)
        r.sigma, r.drift_mode, r.stepper = float(sigma), int(drift_mode), int(stepper)
        return r

    # -- hot path --------------------------------------------------------------------------
    def simulate_ZBC_control_variate(self, n_pairs, seed, path_offset=0, S1=5.0, S2=10.0, K=None,
                                     sigma=0.0, drift_mode=L.HW_DRIFT_BASE,
                                     stepper=L.HW_STEPPER_SUPERPOSED, return_ms=False):
        """Five moment sums {X, Y, X^2, Y^2, XY} over 2*n_pairs samples (float64[5])."""
        r = self._request(n_pairs, seed, path_offset, S1, S2, K, sigma, drift_mode, stepper)
        out = np.zeros(5, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_zbc_moments(self._ctx, C.byref(r), _dp(out), C.byref(ms) if return_ms else None))
        return (out, ms.value) if return_ms else out

    def enqueue_ZBC(self, d_out_ptr, n_pairs, seed, path_offset=0, S1=5.0, S2=10.0, K=None,
                    sigma=0.0, drift_mode=L.HW_DRIFT_BASE, stepper=L.HW_STEPPER_SUPERPOSED,
                    stream=None):
        """Enqueue only: the five sums land at DEVICE address d_out_ptr in stream order."""
        r = self._request(n_pairs, seed, path_offset, S1, S2, K, sigma, drift_mode, stepper)
        L.check(self.lib.hw_zbc_moments_async(self._ctx, C.byref(r), C.c_void_p(d_out_ptr),
                                              C.c_void_p(stream or 0)))

    def simulate_ZBC_runs(self, n_pairs, seed, n_runs=20, seed_stride=12345, path_offset=0, S1=5.0, S2=10.0,
                          K=None, return_ms=False):
        """n_runs independent pricings (seed + k*seed_stride; 2_option:229) in ONE launch -> float64[n_runs, 5]."""
        r = self._request(n_pairs, seed, path_offset, S1, S2, K, 0.0, L.HW_DRIFT_BASE, L.HW_STEPPER_SUPERPOSED)
        out = np.zeros(5 * n_runs, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_zbc_moments_runs(self._ctx, C.byref(r), int(n_runs), int(seed_stride), _dp(out),
                                             C.byref(ms) if return_ms else None))
        out = out.reshape(n_runs, 5)
        return (out, ms.value) if return_ms else out

    def simulate_ZBC_scenarios(self, n_pairs, seed, sigmas, drift_modes, path_offset=0, S1=5.0,
                               S2=10.0, K=None, return_ms=False):
        """len(sigmas) <= 3 scenarios from one pass over the same draws -> float64[n, 5]."""
        n = len(sigmas)
        r = self._request(n_pairs, seed, path_offset, S1, S2, K, 0.0, 0, L.HW_STEPPER_SUPERPOSED)
        sg = (C.c_double * n)(*[float(s) for s in sigmas])
        dm = (C.c_int32 * n)(*[int(m) for m in drift_modes])
        out = np.zeros(5 * n, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_zbc_scenarios(self._ctx, C.byref(r), n, sg, dm, _dp(out), C.byref(ms) if return_ms else None))
        out = out.reshape(n, 5)
        return (out, ms.value) if return_ms else out

    def simulate_zcb(self, n_pairs, seed, path_offset=0, sigma=0.0, return_ms=False):
        """P_sum[n_mat] (float64); P_sum[0] = 2 n_pairs."""
        out = np.zeros(self.params.n_mat, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_zcb_curve(self._ctx, int(n_pairs), int(seed), int(path_offset),
                                      float(sigma), _dp(out), C.byref(ms) if return_ms else None))
        return (out, ms.value) if return_ms else out

    def enqueue_zcb(self, d_out_ptr, n_pairs, seed, path_offset=0, sigma=0.0, stream=None):
        L.check(self.lib.hw_zcb_curve_async(self._ctx, int(n_pairs), int(seed), int(path_offset),
                                            float(sigma), C.c_void_p(d_out_ptr),
                                            C.c_void_p(stream or 0)))

    def compute_average_and_forward(self, P_sum, n_total):
        P_sum = np.ascontiguousarray(P_sum, np.float64)
        P = np.zeros(self.params.n_mat, np.float32)
        f = np.zeros(self.params.n_mat, np.float32)
        L.check(self.lib.hw_average_and_forward(C.byref(self.params), _dp(P_sum), int(n_total),
                                                _fp(P), _fp(f)))
        return P, f

    def simulate_sensitivity(self, n_paths, seed, path_offset=0, S1=5.0, S2=10.0, K=None,
                             antithetic=False, return_ms=False):
        """Sum of the pathwise Vega integrand (term1 - term2) over the paths."""
        out = np.zeros(1, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_vega_pathwise(self._ctx, int(n_paths), int(seed), int(path_offset),
                                          float(S1), float(S2),
                                          float(K_DEFAULT if K is None else K),
                                          1 if antithetic else 0, _dp(out), C.byref(ms) if return_ms else None))
        return (float(out[0]), ms.value) if return_ms else float(out[0])

    def simulate_sensitivity_runs(self, n_paths, seed, n_runs=20, seed_stride=982451653, path_offset=0, S1=5.0,
                                  S2=10.0, K=None, antithetic=False, return_ms=False):
        """n_runs Vega sums in one launch (the bootstrap of 3_sens:527-654): float64[n_runs]."""
        out = np.zeros(n_runs, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_vega_pathwise_runs(self._ctx, int(n_paths), int(seed), int(path_offset), float(S1),
                                               float(S2), float(K_DEFAULT if K is None else K),
                                               int(bool(antithetic)), int(n_runs), int(seed_stride), _dp(out),
                                               C.byref(ms) if return_ms else None))
        return (out, ms.value) if return_ms else out

    def enqueue_sensitivity(self, d_out_ptr, n_paths, seed, path_offset=0, S1=5.0, S2=10.0, K=None,
                            antithetic=False, stream=None):
        L.check(self.lib.hw_vega_pathwise_async(
            self._ctx, int(n_paths), int(seed), int(path_offset), float(S1), float(S2),
            float(K_DEFAULT if K is None else K), 1 if antithetic else 0,
            C.c_void_p(d_out_ptr), C.c_void_p(stream or 0)))

    def run_finite_difference(self, n_pairs, seed, epsilon=0.001, path_offset=0, S1=5.0, S2=10.0,
                              K=None, fused=True):
        """Central-difference Vega with common random numbers (3_sens:400-446).

        fused=True prices sigma-eps and sigma+eps from ONE pass over the draws; fused=False
        launches the ZBC kernel twice with the same seed, as the reference does."""
        s0 = self.params.sigma
        P0S2 = float(self.P[-1])                                   # 3_sens:153-154
        if fused:
            sums = self.simulate_ZBC_scenarios(n_pairs, seed, [s0 - epsilon, s0 + epsilon],
                                               [L.HW_DRIFT_SHIFTED, L.HW_DRIFT_SHIFTED],
                                               path_offset, S1, S2, K)
            sm, sp = sums[0], sums[1]
        else:
            sm = self.simulate_ZBC_control_variate(n_pairs, seed, path_offset, S1, S2, K,
                                                   s0 - epsilon, L.HW_DRIFT_SHIFTED)
            sp = self.simulate_ZBC_control_variate(n_pairs, seed, path_offset, S1, S2, K,
                                                   s0 + epsilon, L.HW_DRIFT_SHIFTED)
        pm = cv_price(sm, 2 * n_pairs, P0S2)["price"]
        pp = cv_price(sp, 2 * n_pairs, P0S2)["price"]
        return {"p_minus": pm, "p_plus": pp, "vega_fd": (pp - pm) / (2.0 * epsilon)}

    def run_finite_difference_recalibrated(self, n_pairs, seed, epsilon=0.001, path_offset=0, S1=5.0,
                                           S2=10.0, K=None):
        """The reference's diagnostic FD (3_sens:449-525): theta held fixed, the market curves
        re-simulated at sigma -+ eps (simulate_zcb + compute_average_and_forward) and the option
        priced on them with the base drift table.  Known to be ~2x off (README.md:51)."""
        s0 = self.params.sigma
        P_keep, f_keep = self.P, self.f
        prices = []
        try:
            for sg in (s0 - epsilon, s0 + epsilon):
                P_sum = self.simulate_zcb(n_pairs, seed, path_offset, sigma=sg)      # recompute_market_data
                Ps, fs = self.compute_average_and_forward(P_sum, 2 * n_pairs)
                self.load_market_data(Ps, fs)
                sums = self.simulate_ZBC_control_variate(n_pairs, seed, path_offset, S1, S2, K, sigma=sg,
                                                         drift_mode=L.HW_DRIFT_BASE)
                prices.append(cv_price(sums, 2 * n_pairs, float(Ps[-1]))["price"])
        finally:
            self.load_market_data(P_keep, f_keep)                                    # 3_sens:516-520
        return {"p_minus": prices[0], "p_plus": prices[1],
                "vega_fd_recalibrated": (prices[1] - prices[0]) / (2.0 * epsilon)}

    def simulate_paths_show(self, n, seed, path_offset=0):
        out = np.zeros((n, self.params.n_steps + 1), np.float32)
        L.check(self.lib.hw_sample_paths(self._ctx, int(n), int(seed), int(path_offset), _fp(out)))
        return out

    def recover_theta(self, f):
        f = np.ascontiguousarray(f, np.float32)
        n = self.params.n_mat
        rec, orig, Ts = (np.zeros(n, np.float32) for _ in range(3))
        L.check(self.lib.hw_recover_theta(C.byref(self.params), _fp(f), _fp(rec), _fp(orig), _fp(Ts)))
        return rec, orig, Ts

    # -- device RNG probes (tests) -----------------------------------------------------------
    def debug_normals(self, seed, subseq, n_blocks):
        out = np.zeros(4 * n_blocks, np.float32)
        L.check(self.lib.hw_debug_normals(self._ctx, int(seed), int(subseq), int(n_blocks), _fp(out)))
        return out.reshape(n_blocks, 4)

    def debug_raw(self, seed, subseq, n_blocks):
        out = np.zeros(4 * n_blocks, np.uint32)
        L.check(self.lib.hw_debug_raw(self._ctx, int(seed), int(subseq), int(n_blocks),
                                      out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out.reshape(n_blocks, 4)

    def debug_exp2(self, x):
        """The double engine's exp2 evaluated on the GPU (hw_debug_exp2)."""
        x = np.ascontiguousarray(x, np.float64)
        out = np.zeros_like(x)
        L.check(self.lib.hw_debug_exp2(self._ctx, _dp(x), int(x.size), _dp(out)))
        return out


class HullWhitePipeline:
    """hw_pipe_*: a stream of independent pricing calls on one GPU with `queues` launches in flight (include/hw_mc.h).
    Every request is one full pricing call; the sums equal those of separate HullWhiteEngine calls bit for bit."""

    def __init__(self, params=None, device=0, queues=2, **kw):
        self.lib = L.load()
        self.params = params if params is not None else default_params(**kw)
        self._pipe = C.c_void_p()
        L.check(self.lib.hw_pipe_create(C.byref(self.params), int(device), int(queues), C.byref(self._pipe)))

    def close(self):
        if getattr(self, "_pipe", None):
            self.lib.hw_pipe_destroy(self._pipe)
            self._pipe = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def load_market_data(self, P, f):
        P = np.ascontiguousarray(P, np.float32)
        f = np.ascontiguousarray(f, np.float32)
        L.check(self.lib.hw_pipe_set_market(self._pipe, _fp(P), _fp(f), len(P)))

    @staticmethod
    def _requests(n_pairs, seeds, path_offset, S1, S2, K, sigma, drift_mode, stepper):
        reqs = (L.HwZbcRequest * len(seeds))()
        for r, seed in zip(reqs, seeds):
            r.n_pairs, r.seed, r.path_offset = int(n_pairs), int(seed), int(path_offset)
            r.S1, r.S2, r.K = float(S1), float(S2), float(K_DEFAULT if K is None else K)
            r.sigma, r.drift_mode, r.stepper = float(sigma), int(drift_mode), int(stepper)
        return reqs

    def queue(self, q):
        """(context of queue q as a borrowed HullWhiteEngine, index of the candidate stream it uses, number of queues):
        hw_pipe_queue (include/hw_mc_diag.h).  The borrowed engine is for the profiling hooks only; closing it is a no-op."""
        ctx, idx, nq = C.c_void_p(), C.c_int32(-1), C.c_int32(0)
        L.check(self.lib.hw_pipe_queue(self._pipe, int(q), C.byref(ctx), C.byref(idx), C.byref(nq)))
        eng = HullWhiteEngine.__new__(HullWhiteEngine)
        eng.lib, eng.params, eng._ctx, eng.P, eng.f = self.lib, self.params, ctx, None, None
        eng.close = lambda: None                        # owned by the pipe
        return eng, idx.value, nq.value

    def tune(self, n_pairs, seed=1, launches=0, **kw):
        """Pick the streams on this workload (untimed by callers); returns the best time per call in ms."""
        reqs = self._requests(n_pairs, [seed], kw.get("path_offset", 0), kw.get("S1", 5.0), kw.get("S2", 10.0), kw.get("K"),
                              kw.get("sigma", 0.0), kw.get("drift_mode", L.HW_DRIFT_BASE), kw.get("stepper", L.HW_STEPPER_SUPERPOSED))
        ms = C.c_float(0)
        L.check(self.lib.hw_pipe_tune(self._pipe, reqs, int(launches), C.byref(ms)))
        return ms.value

    def simulate_ZBC_batch(self, n_pairs, seeds, path_offset=0, S1=5.0, S2=10.0, K=None, sigma=0.0,
                           drift_mode=L.HW_DRIFT_BASE, stepper=L.HW_STEPPER_SUPERPOSED, return_ms=False):
        """One pricing call per seed, alternating over the queues -> float64[len(seeds), 5] (and the wall time in ms)."""
        seeds = list(seeds)
        reqs = self._requests(n_pairs, seeds, path_offset, S1, S2, K, sigma, drift_mode, stepper)
        out = np.zeros((len(seeds), 5), np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_pipe_zbc_moments(self._pipe, reqs, len(seeds), _dp(out), C.byref(ms)))
        return (out, ms.value) if return_ms else out


class HullWhiteFleet:
    """Several engines (one per device) in ONE process behind an RCCL communicator: every call shards the
    paths by global id, launches all devices asynchronously and sums with one grouped ncclAllReduce
    (hw_comm_*, include/hw_mc.h).  bench.py uses one process per GPU with torch.distributed instead
    (parallel.py); this class is what the q1/q2/q3 executables use for --gpus N."""

    def __init__(self, engines, host_sum=False):
        """host_sum=True: no RCCL -- the per-device worker threads bring their sums to the host and the library adds
        them in rank order (HW_COMM_HOST_SUM); the engines may then share a device (rehearsal on a one-GPU box)."""
        self.lib = L.load()
        self.engines = list(engines)
        arr = (C.c_void_p * len(self.engines))(*[e._ctx for e in self.engines])
        self._comm = C.c_void_p()
        L.check(self.lib.hw_comm_create_ex(len(self.engines), arr, L.HW_COMM_HOST_SUM if host_sum else L.HW_COMM_RCCL,
                                           C.byref(self._comm)))

    def close(self):
        if self._comm:
            self.lib.hw_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def simulate_ZBC_control_variate(self, n_pairs, seed, path_offset=0, S1=5.0, S2=10.0, K=None, sigma=0.0,
                                     drift_mode=L.HW_DRIFT_BASE, stepper=L.HW_STEPPER_SUPERPOSED,
                                     return_ms=False):
        r = self.engines[0]._request(n_pairs, seed, path_offset, S1, S2, K, sigma, drift_mode, stepper)
        out = np.zeros(5, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_comm_zbc_moments(self._comm, C.byref(r), _dp(out), C.byref(ms)))
        return (out, ms.value) if return_ms else out

    def simulate_ZBC_runs(self, n_pairs, seed, n_runs=20, seed_stride=12345, path_offset=0, S1=5.0, S2=10.0, K=None,
                          return_ms=False):
        """The 20-run validation over the fleet: one launch per device for all runs, one all-reduce of 5*n_runs doubles."""
        r = self.engines[0]._request(n_pairs, seed, path_offset, S1, S2, K, 0.0, L.HW_DRIFT_BASE, L.HW_STEPPER_SUPERPOSED)
        out = np.zeros(5 * n_runs, np.float64)
        ms = C.c_float(0)
        L.check(self.lib.hw_comm_zbc_moments_runs(self._comm, C.byref(r), int(n_runs), int(seed_stride), _dp(out), C.byref(ms)))
        out = out.reshape(n_runs, 5)
        return (out, ms.value) if return_ms else out

    def simulate_zcb(self, n_pairs, seed, sigma=0.0):
        out = np.zeros(self.engines[0].params.n_mat, np.float64)
        L.check(self.lib.hw_comm_zcb_curve(self._comm, int(n_pairs), int(seed), float(sigma), _dp(out), None))
        return out

    def simulate_sensitivity(self, n_paths, seed, S1=5.0, S2=10.0, K=None, antithetic=False):
        out = np.zeros(1, np.float64)
        L.check(self.lib.hw_comm_vega_pathwise(self._comm, int(n_paths), int(seed), float(S1), float(S2),
                                               float(K_DEFAULT if K is None else K), int(bool(antithetic)),
                                               _dp(out), None))
        return float(out[0])
