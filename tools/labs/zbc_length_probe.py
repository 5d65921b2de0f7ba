#!/usr/bin/env python3
"""ZBC path-kernel time at 2^20 pairs against the number of executed steps (S1 = 2.5, 5, 9.9 -> 250, 500, 990 steps): the
straight line T = t0 + c * steps separates what a launch spends outside the time loop from the per-step rate AT THIS SIZE
(the roof for the 1000-step curve kernel at the same size).    python tools/zbc_length_probe.py"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import hwmc
p = hwmc.default_params("head")
e = hwmc.HullWhiteEngine(p, device=0)
e.load_market_data(*hwmc.analytic_curve(p))
n = 1 << 20
for _ in range(300): e.simulate_ZBC_control_variate(n, 1)
for S1, S2 in ((5.0, 10.0), (9.9, 10.0), (2.5, 10.0)):
    ms = []
    for _ in range(15):
        e.simulate_ZBC_control_variate(n, 1, S1=S1, S2=S2)
        e.profile_enable(1); e.simulate_ZBC_control_variate(n, 1, S1=S1, S2=S2); ms.append(float(e.profile_read(1)[0]))
    print("ZBC S1=%.1f (%d steps): med %.4f ms min %.4f" % (S1, int(S1*100+1e-6), np.median(ms), min(ms)))
