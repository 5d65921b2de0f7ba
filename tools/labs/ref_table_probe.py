#!/usr/bin/env python3
"""REFERENCE-stepper kernel, drift table by wave-uniform scalar loads vs staged in LDS, under the two HIP runtimes a process can
end up with: the system's (/opt/rocm, what bin/q* and a plain `import hwmc` get) and the one bundled with PyTorch (what
bench.py gets, because it imports torch first).    python tools/ref_table_probe.py [--torch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "--torch" in sys.argv:
    import torch  # noqa: F401
    torch.zeros(1, device="cuda")
import hwmc  # noqa: E402

p = hwmc.default_params("head")
e = hwmc.HullWhiteEngine(p, device=0)
e.load_market_data(*hwmc.analytic_curve(p))
n = 1 << 20
for _ in range(300):
    e.simulate_ZBC_control_variate(n, 1)
for lds in (0, 1, 0, 1):
    e.set_launch(table_in_lds=lds)
    ms = []
    for i in range(25):
        e.simulate_ZBC_control_variate(n, 1, stepper=hwmc.HW_STEPPER_REFERENCE)
        e.profile_enable(1)
        e.simulate_ZBC_control_variate(n, 1, stepper=hwmc.HW_STEPPER_REFERENCE)
        ms.append(float(e.profile_read(1)[0]))
    print("%-18s table %-12s med %.4f ms  min %.4f" % ("torch runtime" if "--torch" in sys.argv else "system runtime", "in LDS" if lds else "scalar loads", np.median(ms), min(ms)))
