#!/usr/bin/env python3
"""One balanced round or several ragged ones?  The ZBC path kernel at 2^20 pairs, alone on the chip, for launch shapes that
give every SIMD the same number of equally long waves from the start (grid capped at a whole number of workgroups per
CU, the threads looping) against the default (one slot per thread: the dispatcher hands workgroups out as CUs free up).
hipEvent pair around the path kernel, back-to-back launches of each shape, interleaved rounds.
    python tools/round_shape_probe.py [log2 pairs = 20]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
p = hwmc.default_params("head")
e = hwmc.HullWhiteEngine(p, device=0)
P, f = hwmc.analytic_curve(p)
e.load_market_data(P, f)
shapes = [("default (ilp auto, grid = all the work)", -1, 0, -1),
          ("ilp 1, grid = all the work", -1, 0, 1),
          ("ilp 2, grid = all the work", -1, 0, 2)]
for block in (256, 512, 1024):
    for ilp in (1, 2):
        for per_cu in (1, 2, 3, 4, 6, 8):
            waves = per_cu * block // 64 // 4
            if waves > (8 if ilp == 1 else 6) or waves < 2:
                continue
            shapes.append(("block %4d ilp %d, %d workgroups per CU (%d waves/SIMD)" % (block, ilp, per_cu, waves), block, per_cu, ilp))
times = {s[0]: [] for s in shapes}


def run(s, reps):
    name, block, cap, ilp = s
    e.set_launch(block, cap, 0, ilp)
    out = []
    for _ in range(3):
        e.simulate_ZBC_control_variate(n, 7)
    for r in range(reps):
        e.profile_enable(1)
        e.simulate_ZBC_control_variate(n, 1234 + r)
        out.append(float(e.profile_read(1)[0]))
    return out


import time
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    e.simulate_ZBC_control_variate(n, 7)
for rnd in range(5):
    for s in shapes:
        times[s[0]] += run(s, 8)
print("ZBC path kernel alone on the chip, 2^%d pairs x 500 steps: median / min ms over %d launches; geometry" % (int(np.log2(n)), 40))
for s in shapes:
    e.set_launch(s[1], s[2], 0, s[3])
    g = e.launch_geometry("zbc", n)
    t = np.array(times[s[0]])
    print("%-58s %.4f  %.4f   grid %5d x %4d threads, %d iteration(s)" % (s[0], np.median(t), t.min(), g["grid"], g["block"], g["iters"]))
e.close()
