#!/usr/bin/env python3
"""Pairs per thread (P) of the blocked curve kernel against batch size, grid and precision (HWMC_CURVE_P forces P): where the
engine's rule (launch_zcb, csrc/hw_engine.hip) should switch.  python tools/labs/curve_p_now.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402

for steps, prec, sizes in ((1000, "f32", (20, 21, 22, 23, 24, 26)), (4000, "f32", (20, 22, 24, 26)), (4000, "f64", (22, 26)), (1000, "f64", (20, 22, 24))):
    for lg in sizes:
        row = []
        for P in ("1", "2", "4"):
            os.environ["HWMC_CURVE_P"] = P
            p = hwmc.default_params("head", n_steps=steps, precision=prec)
            with hwmc.HullWhiteEngine(p, device=0) as e:
                n = 1 << lg
                reps = 12 if lg <= 24 and steps == 1000 else 3
                for _ in range(max(2, (100 >> max(0, lg - 20)) * (1 if steps == 1000 else 0) + 2)):
                    e.simulate_zcb(n, 1)
                ms = []
                for j in range(reps):
                    e.simulate_zcb(n, 5 + j)
                    e.profile_enable(1)
                    e.simulate_zcb(n, 5 + j)
                    ms.append(float(e.profile_read(1)[0]))
                row.append(float(np.median(ms)))
        best = int(np.argmin(row))
        print("%d-step grid %s 2^%d pairs:  P=1 %.4f  P=2 %.4f  P=4 %.4f ms   best P=%d (P=4 / best %.4f, P=2 / best %.4f)"
              % (steps, prec, lg, row[0], row[1], row[2], (1, 2, 4)[best], row[2] / row[best], row[1] / row[best]), flush=True)
