#!/usr/bin/env python3
"""Host-side cost of the per-device worker threads of hw_comm_*: wall time per pricing call of a tiny batch (the kernel is
~25 us, so the call is dominated by launch + synchronisation + thread hand-over) for a single context called directly and
for HW_COMM_HOST_SUM fleets of 1, 2, 4 contexts that all sit on device 0 (the only way to run G > 1 on a one-GPU box: the
kernels of the G shards then share one GPU, so only the HOST side of the numbers means anything).
    python tools/comm_dispatch_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hwmc  # noqa: E402

p = hwmc.default_params("head")
P, f = hwmc.analytic_curve(p)
engines = [hwmc.HullWhiteEngine(p, device=0) for _ in range(4)]
for e in engines:
    e.load_market_data(P, f)
n, reps = 1 << 12, 2000


def timeit(fn):
    for _ in range(200):
        fn()
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t.append((time.perf_counter() - t0) / reps * 1e6)
    return float(np.median(t)), float(min(t))


print("%-44s %10s %10s" % ("2^12 pairs per call, wall us per call", "median", "min"))
print("%-44s %10.1f %10.1f" % ("one context, hw_zbc_moments (no threads)", *timeit(lambda: engines[0].simulate_ZBC_control_variate(n, 1))))
for G in (1, 2, 4):
    with hwmc.HullWhiteFleet(engines[:G], host_sum=True) as fleet:
        print("%-44s %10.1f %10.1f" % ("fleet of %d, worker threads, host sum" % G, *timeit(lambda: fleet.simulate_ZBC_control_variate(n, 1))))
        pauses = []
        for _ in range(50):                              # a call after a 2 ms pause: the workers have gone to sleep
            time.sleep(0.002)
            t0 = time.perf_counter()
            fleet.simulate_ZBC_control_variate(n, 1)
            pauses.append((time.perf_counter() - t0) * 1e6)
        print("%-44s %10.1f %10.1f" % ("   ... first call after a 2 ms pause", float(np.median(pauses)), min(pauses)))
for e in engines:
    e.close()
