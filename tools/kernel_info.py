#!/usr/bin/env python3
"""Register, LDS and occupancy report of the path kernels (what the reference prints with cudaFuncGetAttributes,
src/3_sensitivity_analysis.cu:183-239)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hwmc  # noqa: E402

for prec in ("f32", "f64"):
    e = hwmc.HullWhiteEngine(hwmc.default_params("head", precision=prec), device=0)
    for k in ("zbc", "zbc_ref", "zcb", "zcb_blocked", "vega"):
        try:
            print(prec, "%-12s" % k, e.kernel_info(k))
        except hwmc.HwError as err:
            print(prec, k, "n/a:", err)
    e.close()
