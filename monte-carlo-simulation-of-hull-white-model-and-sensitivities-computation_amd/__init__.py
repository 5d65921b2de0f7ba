"""MI355X-native Hull-White one-factor Monte Carlo hot path (gfx950 HIP kernels behind a C ABI).

The directory name contains hyphens (it mirrors the reference repository's name), so import it
through the `hwmc` shim at the repository root:  `import hwmc`.
"""
from . import _lib
from ._lib import (HW_DRIFT_BASE, HW_DRIFT_SHIFTED, HW_F32, HW_F64, HW_STEPPER_REFERENCE,
                   HW_STEPPER_SUPERPOSED, HW_TABLES_F64, HW_TABLES_REF_F32, HwError)
from . import parallel
from .build import HIPCC_FLAGS, build_library
from .engine import (K_DEFAULT, HullWhiteEngine, HullWhiteFleet, HullWhitePipeline, analytic_curve, closed_form_vega,
                     closed_form_zbc, curve_standard_errors, cv_price, default_params, run_statistics)

__all__ = ["parallel", "HullWhiteEngine", "HullWhiteFleet", "HullWhitePipeline", "analytic_curve", "cv_price", "default_params", "run_statistics",
           "closed_form_zbc", "closed_form_vega", "curve_standard_errors",
           "build_library", "HIPCC_FLAGS", "HwError", "K_DEFAULT", "HW_F32", "HW_F64", "HW_TABLES_F64",
           "HW_TABLES_REF_F32", "HW_STEPPER_SUPERPOSED", "HW_STEPPER_REFERENCE", "HW_DRIFT_BASE",
           "HW_DRIFT_SHIFTED"]
