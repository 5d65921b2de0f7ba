"""ctypes binding of libhw_hip.so (the C ABI declared in include/hw_mc.h).

The library is built in-tree by build.py (hipcc --offload-arch=gfx950).  Loading fails loudly
when the shared object is missing: there is no Python or CPU fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HWMC_LIB") or os.path.join(_HERE, "libhw_hip.so")   # HWMC_LIB: A/B builds (tools/)

HW_OK = 0
HW_E_INVALID, HW_E_HIP, HW_E_NO_MARKET, HW_E_NOMEM = -1, -2, -3, -4
HW_F32, HW_F64 = 0, 1
HW_TABLES_F64, HW_TABLES_REF_F32 = 0, 1
HW_STEPPER_SUPERPOSED, HW_STEPPER_REFERENCE = 0, 1
HW_DRIFT_BASE, HW_DRIFT_SHIFTED = 0, 1
HW_COMM_RCCL, HW_COMM_HOST_SUM = 0, 1
ABI_VERSION = 6                 # HW_ABI_VERSION of include/hw_mc.h


class HwParams(C.Structure):
    _fields_ = [("a", C.c_double), ("sigma", C.c_double), ("r0", C.c_double),
                ("T_final", C.c_double), ("n_steps", C.c_int32), ("n_mat", C.c_int32),
                ("th_break", C.c_double), ("th_a1", C.c_double), ("th_b1", C.c_double),
                ("th_a2", C.c_double), ("th_b2", C.c_double),
                ("precision", C.c_int32), ("table_mode", C.c_int32)]


class HwZbcRequest(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("seed", C.c_uint64), ("path_offset", C.c_uint64),
                ("S1", C.c_double), ("S2", C.c_double), ("K", C.c_double),
                ("sigma", C.c_double), ("drift_mode", C.c_int32), ("stepper", C.c_int32)]


class HwDeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("device", C.c_int32), ("compute_units", C.c_int32),
                ("clock_mhz", C.c_int32), ("wavefront_size", C.c_int32),
                ("lds_per_cu_bytes", C.c_int64), ("hbm_bytes", C.c_int64)]


class HwKernelInfo(C.Structure):
    _fields_ = [("vgprs", C.c_int32), ("sgprs", C.c_int32), ("lds_bytes", C.c_int32),
                ("scratch_bytes", C.c_int32), ("max_threads_per_block", C.c_int32),
                ("block_size", C.c_int32), ("blocks_per_cu", C.c_int32)]


_DP = C.POINTER(C.c_double)
_FP = C.POINTER(C.c_float)
_CTX = C.c_void_p

# name -> (restype, argtypes); every symbol include/hw_mc.h declares (the stable boundary)
SIGNATURES = {
    "hw_abi_version": (C.c_int, []),
    "hw_last_error": (C.c_char_p, []),
    "hw_default_params": (C.c_int, [C.POINTER(HwParams), C.c_char_p]),
    "hw_create": (C.c_int, [C.POINTER(HwParams), C.c_int, C.POINTER(_CTX)]),
    "hw_destroy": (None, [_CTX]),
    "hw_get_device_info": (C.c_int, [_CTX, C.POINTER(HwDeviceInfo)]),
    "hw_get_params": (C.c_int, [_CTX, C.POINTER(HwParams)]),
    "hw_set_market": (C.c_int, [_CTX, _FP, _FP, C.c_int32]),
    "hw_zbc_moments": (C.c_int, [_CTX, C.POINTER(HwZbcRequest), _DP, _FP]),
    "hw_zbc_moments_async": (C.c_int, [_CTX, C.POINTER(HwZbcRequest), C.c_void_p, C.c_void_p]),
    "hw_zbc_moments_runs": (C.c_int, [_CTX, C.POINTER(HwZbcRequest), C.c_int32, C.c_uint64, _DP, _FP]),
    "hw_zbc_scenarios": (C.c_int, [_CTX, C.POINTER(HwZbcRequest), C.c_int32, _DP,
                                   C.POINTER(C.c_int32), _DP, _FP]),
    "hw_zcb_curve": (C.c_int, [_CTX, C.c_int64, C.c_uint64, C.c_uint64, C.c_double, _DP, _FP]),
    "hw_zcb_curve_async": (C.c_int, [_CTX, C.c_int64, C.c_uint64, C.c_uint64, C.c_double,
                                     C.c_void_p, C.c_void_p]),
    "hw_vega_pathwise": (C.c_int, [_CTX, C.c_int64, C.c_uint64, C.c_uint64, C.c_double,
                                   C.c_double, C.c_double, C.c_int32, _DP, _FP]),
    "hw_vega_pathwise_runs": (C.c_int, [_CTX, C.c_int64, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double,
                                        C.c_int32, C.c_int32, C.c_uint64, _DP, C.POINTER(C.c_float)]),
    "hw_vega_pathwise_async": (C.c_int, [_CTX, C.c_int64, C.c_uint64, C.c_uint64, C.c_double,
                                         C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]),
    "hw_sample_paths": (C.c_int, [_CTX, C.c_int32, C.c_uint64, C.c_uint64, _FP]),
    "hw_comm_create": (C.c_int, [C.c_int32, C.POINTER(_CTX), C.POINTER(C.c_void_p)]),
    "hw_comm_create_ex": (C.c_int, [C.c_int32, C.POINTER(_CTX), C.c_int32, C.POINTER(C.c_void_p)]),
    "hw_comm_destroy": (None, [C.c_void_p]),
    "hw_comm_zbc_moments": (C.c_int, [C.c_void_p, C.POINTER(HwZbcRequest), _DP, C.POINTER(C.c_float)]),
    "hw_comm_zbc_moments_runs": (C.c_int, [C.c_void_p, C.POINTER(HwZbcRequest), C.c_int32, C.c_uint64, _DP, C.POINTER(C.c_float)]),
    "hw_comm_zcb_curve": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, _DP, C.POINTER(C.c_float)]),
    "hw_comm_vega_pathwise": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_double,
                                        C.c_int32, _DP, C.POINTER(C.c_float)]),
    "hw_pipe_create": (C.c_int, [C.POINTER(HwParams), C.c_int, C.c_int32, C.POINTER(C.c_void_p)]),
    "hw_pipe_destroy": (None, [C.c_void_p]),
    "hw_pipe_set_market": (C.c_int, [C.c_void_p, _FP, _FP, C.c_int32]),
    "hw_pipe_tune": (C.c_int, [C.c_void_p, C.POINTER(HwZbcRequest), C.c_int32, C.POINTER(C.c_float)]),
    "hw_pipe_zbc_moments": (C.c_int, [C.c_void_p, C.POINTER(HwZbcRequest), C.c_int32, _DP, C.POINTER(C.c_float)]),
    "hw_drift_tables": (C.c_int, [_CTX, C.c_double, C.c_int32, _DP, _DP]),
    "hw_average_and_forward": (C.c_int, [C.POINTER(HwParams), _DP, C.c_int64, _FP, _FP]),
    "hw_recover_theta": (C.c_int, [C.POINTER(HwParams), _FP, _FP, _FP, _FP]),
    "hw_cv_price": (C.c_int, [_DP, C.c_int64, C.c_double, _DP]),
    "hw_analytic_curve": (C.c_int, [C.POINTER(HwParams), _FP, _FP]),
}

# ... and every symbol include/hw_mc_diag.h declares (launch knobs, kernel resources, device probes, profiling hooks)
DIAG_SIGNATURES = {
    "hw_set_launch": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "hw_debug_normals": (C.c_int, [_CTX, C.c_uint64, C.c_uint64, C.c_int32, _FP]),
    "hw_debug_raw": (C.c_int, [_CTX, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_uint32)]),
    "hw_debug_exp2": (C.c_int, [_CTX, _DP, C.c_int32, _DP]),
    "hw_profile_enable": (C.c_int, [_CTX, C.c_int32]),
    "hw_profile_read": (C.c_int, [_CTX, _FP, C.c_int32, C.POINTER(C.c_int32)]),
    "hw_profile_clock": (C.c_int, [_CTX, _DP]),
    "hw_profile_clock_samples": (C.c_int, [_CTX, _DP, _DP, _DP, C.c_int32, C.POINTER(C.c_int32)]),
    "hw_profile_spans": (C.c_int, [_CTX, _CTX, _FP, _FP, C.c_int32, C.POINTER(C.c_int32)]),
    "hw_get_launch_geometry": (C.c_int, [_CTX, C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "hw_get_kernel_info": (C.c_int, [_CTX, C.c_char_p, C.POINTER(HwKernelInfo)]),
    "hw_pipe_queue": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(_CTX), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_lib = None


class HwError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("hw_mc error %d: %s" % (code, text))
        self.code = code


def load():
    """dlopen libhw_hip.so and attach the signatures.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP engine has not been built (run __graft_entry__.build() or "
            "`make -C <package>/csrc`).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.hw_abi_version() != ABI_VERSION:
        raise ImportError("libhw_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise HwError(rc, (load().hw_last_error() or b"").decode("utf-8", "replace"))
