"""ctypes binding of the CPU oracle (oracle/libhw_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.  See oracle/hw_oracle.h for
what the oracle restates (reference file:line per function) and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhw_oracle.so")


class OraModel(C.Structure):
    _fields_ = [("a", C.c_double), ("sigma", C.c_double), ("r0", C.c_double),
                ("T_final", C.c_double), ("n_steps", C.c_int32), ("n_mat", C.c_int32),
                ("th_break", C.c_double), ("th_a1", C.c_double), ("th_b1", C.c_double),
                ("th_a2", C.c_double), ("th_b2", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(_LIB_PATH)
            for s in ("hw_oracle.c", "hw_oracle_body.inc", "hw_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhw_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ora_max_threads.restype = C.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def usable_cores(omp_max):
    """Threads worth starting: the smallest of OpenMP's default, the affinity mask and the cgroup CPU quota
    (a GPU box hands a job a share of the host's cores; more threads than that only get throttled)."""
    n, how = omp_max, "omp_get_max_threads"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, how = aff, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    period = float(fh.read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    n, how = q, "cgroup cpu quota"
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, how



class Oracle:
    """Thin object wrapper: one model, float32 ("f32", the reference's arithmetic) or float64."""

    def __init__(self, profile="head", n_steps=1000, n_mat=101, sigma=None, precision="f32"):
        self.l = lib()
        self.m = OraModel()
        rc = self.l.ora_default_model(C.byref(self.m), profile.encode())
        if rc != 0:
            raise ValueError("unknown theta profile %r" % profile)
        self.m.n_steps = n_steps
        self.m.n_mat = n_mat
        if sigma is not None:
            self.m.sigma = sigma
        assert precision in ("f32", "f64")
        self.precision = precision
        self._rt = np.float32 if precision == "f32" else np.float64

    # -- helpers ---------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self.l, "%s_%s" % (name, self.precision))

    def _rp(self, a):
        return _fp(a) if self.precision == "f32" else _dp(a)

    @property
    def sigma(self):
        return self.m.sigma

    def omp_max_threads(self):
        return self.l.ora_max_threads()

    def max_threads(self):
        """Default thread count of every call below: what the job may actually use (see usable_cores)."""
        return usable_cores(self.l.ora_max_threads())[0]

    def steps_to(self, S1):
        return self.l.ora_steps_to(C.byref(self.m), C.c_double(S1))

    def constants(self, sigma=None):
        out = np.zeros(6, self._rt)
        self._fn("ora_constants")(C.byref(self.m), C.c_double(sigma or self.m.sigma), self._rp(out))
        return out

    def tables(self, sigma=None):
        n = self.m.n_steps
        d = np.zeros(n, self._rt)
        sd = np.zeros(n, self._rt)
        self._fn("ora_tables")(C.byref(self.m), C.c_double(sigma or self.m.sigma),
                               self._rp(d), self._rp(sd))
        return d, sd

    def shifted_drift(self, sigma_new, sigma_old=None):
        d = np.zeros(self.m.n_steps, self._rt)
        self._fn("ora_shifted_drift")(C.byref(self.m), C.c_double(sigma_new),
                                      C.c_double(sigma_old or self.m.sigma), self._rp(d))
        return d

    def bond_AB(self, P, f, S1, S2, sigma=None):
        out = np.zeros(2, self._rt)
        P = np.ascontiguousarray(P, np.float32)
        f = np.ascontiguousarray(f, np.float32)
        self._fn("ora_bond_AB")(C.byref(self.m), C.c_double(sigma or self.m.sigma), _fp(P), _fp(f),
                                C.c_double(S1), C.c_double(S2), self._rp(out))
        return out

    # -- RNG -------------------------------------------------------------------------------
    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.l.ora_philox4x32_10(c, k, o)
        return [int(x) for x in o]

    def uint4(self, seed, subseq, block):
        o = (C.c_uint32 * 4)()
        self.l.ora_uint4(C.c_uint64(seed), C.c_uint64(subseq), C.c_uint64(block), o)
        return [int(x) for x in o]

    def normal4(self, seed, subseq, block):
        o = (C.c_float * 4)()
        self.l.ora_normal4(C.c_uint64(seed), C.c_uint64(subseq), C.c_uint64(block), o)
        return np.array(list(o), np.float32)

    # -- hot path --------------------------------------------------------------------------
    def zbc_moments(self, P, f, n_pairs, seed, path_offset=0, S1=5.0, S2=10.0, K=None,
                    sigma=None, drift=None, n_threads=None):
        K = float(np.exp(np.float32(-0.1))) if K is None else K
        sigma = sigma or self.m.sigma
        if drift is None:
            drift = self.tables(sigma)[0]
        drift = np.ascontiguousarray(drift, self._rt)
        P = np.ascontiguousarray(P, np.float32)
        f = np.ascontiguousarray(f, np.float32)
        out = np.zeros(5, np.float64)
        rc = self._fn("ora_zbc_moments")(
            C.byref(self.m), C.c_double(sigma), self._rp(drift), _fp(P), _fp(f),
            C.c_int64(n_pairs), C.c_uint64(seed), C.c_uint64(path_offset),
            C.c_double(S1), C.c_double(S2), C.c_double(K), _dp(out),
            C.c_int(n_threads or self.max_threads()))
        if rc != 0:
            raise RuntimeError("ora_zbc_moments rc=%d" % rc)
        return out

    def zcb_curve(self, n_pairs, seed, path_offset=0, sigma=None, drift=None, n_threads=None):
        sigma = sigma or self.m.sigma
        if drift is None:
            drift = self.tables(sigma)[0]
        drift = np.ascontiguousarray(drift, self._rt)
        out = np.zeros(self.m.n_mat, np.float64)
        rc = self._fn("ora_zcb_curve")(
            C.byref(self.m), C.c_double(sigma), self._rp(drift),
            C.c_int64(n_pairs), C.c_uint64(seed), C.c_uint64(path_offset), _dp(out),
            C.c_int(n_threads or self.max_threads()))
        if rc != 0:
            raise RuntimeError("ora_zcb_curve rc=%d" % rc)
        return out

    def average_and_forward(self, P_sum, n_total):
        P_sum = np.ascontiguousarray(P_sum, np.float64)
        P = np.zeros(self.m.n_mat, np.float32)
        f = np.zeros(self.m.n_mat, np.float32)
        self.l.ora_average_and_forward_f32(C.byref(self.m), _dp(P_sum), C.c_int64(n_total),
                                           _fp(P), _fp(f))
        return P, f

    def vega_pathwise(self, P, f, n_paths, seed, path_offset=0, S1=5.0, S2=10.0, K=None,
                      sigma=None, n_threads=None):
        K = float(np.exp(np.float32(-0.1))) if K is None else K
        sigma = sigma or self.m.sigma
        drift, sdrift = self.tables(sigma)
        P = np.ascontiguousarray(P, np.float32)
        f = np.ascontiguousarray(f, np.float32)
        out = np.zeros(1, np.float64)
        rc = self._fn("ora_vega_pathwise")(
            C.byref(self.m), C.c_double(sigma), self._rp(drift), self._rp(sdrift), _fp(P), _fp(f),
            C.c_int64(n_paths), C.c_uint64(seed), C.c_uint64(path_offset),
            C.c_double(S1), C.c_double(S2), C.c_double(K), _dp(out),
            C.c_int(n_threads or self.max_threads()))
        if rc != 0:
            raise RuntimeError("ora_vega_pathwise rc=%d" % rc)
        return float(out[0])

    def vega_pathwise_terms(self, P, f, n_paths, seed, path_offset=0, S1=5.0, S2=10.0, K=None,
                            sigma=None, n_threads=None):
        """(sum(term1 - term2), sum |term1|, sum |term2|): the Vega sum and the scale of its signed terms."""
        K = float(np.exp(np.float32(-0.1))) if K is None else K
        sigma = sigma or self.m.sigma
        drift, sdrift = self.tables(sigma)
        P = np.ascontiguousarray(P, np.float32)
        f = np.ascontiguousarray(f, np.float32)
        out = np.zeros(3, np.float64)
        rc = self._fn("ora_vega_pathwise_terms")(
            C.byref(self.m), C.c_double(sigma), self._rp(drift), self._rp(sdrift), _fp(P), _fp(f),
            C.c_int64(n_paths), C.c_uint64(seed), C.c_uint64(path_offset),
            C.c_double(S1), C.c_double(S2), C.c_double(K), _dp(out),
            C.c_int(n_threads or self.max_threads()))
        if rc != 0:
            raise RuntimeError("ora_vega_pathwise_terms rc=%d" % rc)
        return float(out[0]), float(out[1]), float(out[2])

    def sample_paths(self, n, seed, path_offset=0, sigma=None):
        sigma = sigma or self.m.sigma
        drift = np.zeros(self.m.n_steps, np.float32)
        self.l.ora_tables_f32(C.byref(self.m), C.c_double(sigma), _fp(drift), None)
        out = np.zeros((n, self.m.n_steps + 1), np.float32)
        rc = self.l.ora_sample_paths_f32(C.byref(self.m), C.c_double(sigma), _fp(drift),
                                         C.c_int(n), C.c_uint64(seed), C.c_uint64(path_offset),
                                         _fp(out))
        if rc != 0:
            raise RuntimeError("ora_sample_paths rc=%d" % rc)
        return out

    # -- host post-processing --------------------------------------------------------------
    def cv_price(self, sums, n_total, P0S2):
        sums = np.ascontiguousarray(sums, np.float64)
        res = np.zeros(9, self._rt)
        self._fn("ora_cv_price")(_dp(sums), C.c_int64(n_total), C.c_double(P0S2), self._rp(res))
        keys = ("mean_X", "mean_Y", "var_Y", "cov", "beta", "adjustment", "price", "var_X", "rho")
        return dict(zip(keys, (float(v) for v in res)))

    def recover_theta(self, f):
        f = np.ascontiguousarray(f, np.float32)
        n = self.m.n_mat
        rec, orig, Ts = (np.zeros(n, np.float32) for _ in range(3))
        self.l.ora_recover_theta_f32(C.byref(self.m), _fp(f), _fp(rec), _fp(orig), _fp(Ts))
        return rec, orig, Ts

    def run_statistics(self, x):
        x = np.ascontiguousarray(x, np.float32)
        res = np.zeros(8, np.float32)
        self.l.ora_run_statistics_f32(_fp(x), C.c_int(len(x)), _fp(res))
        keys = ("mean", "variance", "sd", "se", "margin", "ci_lo", "ci_hi", "cv_percent")
        return dict(zip(keys, (float(v) for v in res)))
