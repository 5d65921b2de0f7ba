import sys, time, numpy as np
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc
p = hwmc.default_params("head")
e = hwmc.HullWhiteEngine(p, device=0)
e.load_market_data(*hwmc.analytic_curve(p))
def run(n, S1):
    e.profile_enable(1)
    e.simulate_ZBC_control_variate(n, 1234, S1=S1)
    return float(e.profile_read(1)[0])
t_end = time.time() + 0.5
while time.time() < t_end: run(1 << 20, 5.0)
for S1 in (0.04, 0.2, 1.0, 5.0):
    for lg in (17, 20, 22):
        ts = []
        for _ in range(9):
            run(1 << lg, S1); ts.append(run(1 << lg, S1))
        print(f"S1={S1:5.2f} steps={int(S1*100):4d} pairs=2^{lg}: med {np.median(ts)*1e3:8.1f} us  min {min(ts)*1e3:8.1f} us")
