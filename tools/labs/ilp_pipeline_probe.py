#!/usr/bin/env python3
"""From which batch size do two pairs per thread pay?  an earlier probe measured one launch at a time; this probe
measures the loop bench.py times -- K = 20 and K = 200 pricing calls per region alternating over the best stream pair,
and K = 20 on one queue -- with pairs_in_flight = 1 | 2 forced.  effective_ilp()'s threshold (two chip-fills of pairs)
comes from here: 2^18 loses, 2^19 is mixed, 2^20 and 2^21 win in both modes.
    python tools/ilp_pipeline_probe.py [log2 pairs = 20]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import hwmc  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
p = hwmc.default_params("head")
P, f = hwmc.analytic_curve(p)
engines = []
for _ in range(2):
    e = hwmc.HullWhiteEngine(p, device=0)
    e.load_market_data(P, f)
    engines.append(e)
out = torch.zeros((512, 5), dtype=torch.float64, device=dev)
cands = [torch.cuda.Stream(device=dev) for _ in range(8)]


def region(streams, K):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        k = i % len(streams)
        engines[k].enqueue_ZBC(out[i % 512].data_ptr(), n, 1234 + i, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    region(cands[:1], 16)
best = (1e9, None)
for i in range(8):
    for j in range(i + 1, 8):
        t = min(region([cands[i], cands[j]], 16) for _ in range(2))
        if t < best[0]:
            best = (t, (i, j))
streams = [cands[best[1][0]], cands[best[1][1]]]
print("2^%d pairs per call, stream pair %s" % (int(np.log2(n)), best[1]))
for rep in range(2):
    for ilp in (1, 2):
        for e in engines:
            e.set_launch(pairs_in_flight=ilp)
        t_end = time.perf_counter() + 0.15          # a changed kernel: let the power management settle
        while time.perf_counter() < t_end:
            region(streams, 16)
        t20 = float(np.median([region(streams, 20) for _ in range(15)]))
        t200 = float(np.median([region(streams, 200) for _ in range(5)]))
        t1 = float(np.median([region(streams[:1], 20) for _ in range(9)]))
        print("pairs per thread %d: two queues K = 20 %.4f ms (%.0f M paths/s)   K = 200 %.4f ms (%.0f)   one queue K = 20 %.4f ms (%.0f)"
              % (ilp, t20, 2 * n / t20 / 1e3, t200, 2 * n / t200 / 1e3, t1, 2 * n / t1 / 1e3), flush=True)
