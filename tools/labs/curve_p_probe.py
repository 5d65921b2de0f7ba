#!/usr/bin/env python3
"""Pairs per thread (P) of the blocked curve kernel: kernel time at 2^22 and 2^24 pairs for P = 1, 2, 4
(HWMC_CURVE_P overrides the library's choice; one process per value because it is read at launch time)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import hwmc
e = hwmc.HullWhiteEngine(hwmc.default_params("head"), device=0)
for lg in (22, 24):
    n = 1 << lg
    for _ in range(3): e.simulate_zcb(n, 1)
    ts = []
    for k in range(7):
        e.profile_enable(1); e.simulate_zcb(n, 2 + k); ts.append(float(e.profile_read(1)[0]))
    print("P=%%s 2^%%d: med %%.4f ms  %%.1f M paths/s" %% (sys.argv[1], lg, np.median(ts), 2 * n / np.median(ts) / 1e3))
''' % ROOT
for P in ("1", "2", "4"):
    env = dict(os.environ, HWMC_CURVE_P=P)
    r = subprocess.run([sys.executable, "-c", CHILD, P], env=env, capture_output=True, text=True, timeout=200)
    print(r.stdout.strip() or r.stderr[-500:])
