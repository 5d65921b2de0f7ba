#!/usr/bin/env python3
"""`make run-benchmark`: the reference's optional reduction-strategy benchmark
(src/benchmark_reductions.cu: naive atomics / LDS tree / warp shuffle, 2 warm-ups + 5 timed runs)
re-imagined for the choices that exist on MI355X: how the grid-wide moment sums are finished and
which stepper arithmetic runs.  Writes data/benchmark_reductions.json in the reference's schema
(benchmark_reductions.cu:172-197) so the reference's analyze.py plots it."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, sys, numpy as np
sys.path.insert(0, %r)
import hwmc
p = hwmc.default_params("head")
e = hwmc.HullWhiteEngine(p, device=0)
P, f = hwmc.analytic_curve(p)
e.load_market_data(P, f)
n = 1 << 20
st = hwmc.HW_STEPPER_REFERENCE if sys.argv[1] == "reference" else hwmc.HW_STEPPER_SUPERPOSED
for _ in range(2):
    e.simulate_ZBC_control_variate(n, 1, stepper=st)
ms = []
for k in range(5):
    s, t = e.simulate_ZBC_control_variate(n, 100 + k, stepper=st, return_ms=True)
    ms.append(t)
print(json.dumps({"time_ms": float(np.mean(ms)), "price": hwmc.cv_price(s, 2 * n, float(P[-1]))["price"]}))
''' % ROOT


def main():
    variants = [("two-launch column pass (default)", {"HWMC_SINGLE_LAUNCH": "0"}, "superposed"),
                ("single launch, sharded tickets", {"HWMC_SINGLE_LAUNCH": "1"}, "superposed"),
                ("reference stepper, two-launch", {"HWMC_SINGLE_LAUNCH": "0"}, "reference")]
    results = []
    for name, env, stepper in variants:
        out = subprocess.run([sys.executable, "-c", CHILD, stepper], env={**os.environ, **env}, capture_output=True,
                             text=True, check=True)
        r = json.loads(out.stdout.strip().splitlines()[-1])
        r["method"] = name
        r["throughput_Mpaths_per_sec"] = (2 * (1 << 20) / r["time_ms"]) / 1000.0
        results.append(r)
        print("%-36s %8.3f ms %10.2f M paths/s  price %.8f" % (name, r["time_ms"], r["throughput_Mpaths_per_sec"], r["price"]))
    os.makedirs("data", exist_ok=True)
    with open("data/benchmark_reductions.json", "w") as fh:
        json.dump({"benchmark": "Reduction Methods Performance",
                   "parameters": {"N_PATHS": 1 << 20, "NTPB": 256, "NB": 2048, "S1": 5.0, "S2": 10.0,
                                  "K": float(np.exp(np.float32(-0.1)))},
                   "results": [{"method": r["method"], "time_ms": round(r["time_ms"], 3),
                                "throughput_Mpaths_per_sec": round(r["throughput_Mpaths_per_sec"], 2),
                                "price": r["price"]} for r in results]}, fh, indent=2)
    print("\nSaved data/benchmark_reductions.json")


if __name__ == "__main__":
    main()
