"""Error statistics of the double engine's exp2 (hw_debug_exp2) against libm, on the GPU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc
hwmc.build_library()
p = hwmc.default_params("head", precision="f64")
with hwmc.HullWhiteEngine(p, device=0) as e:
    rng = np.random.default_rng(7)
    for name, x in (("uniform(-3,3)", rng.uniform(-3, 3, 1 << 22)), ("normal(0,0.2)", rng.normal(0, 0.2, 1 << 22)),
                    ("uniform(-40,40)", rng.uniform(-40, 40, 1 << 22))):
        rel = e.debug_exp2(x) / np.exp2(x) - 1.0
        print("%-16s n=%d max|rel|=%.3e mean=%+.3e sd=%.3e mean/se=%+.2f" % (name, x.size, np.abs(rel).max(), rel.mean(), rel.std(),
                                                                          rel.mean() / (rel.std() / np.sqrt(x.size))))
