#!/usr/bin/env python3
"""Does a priority difference between the two queues help the pipelined pricing loop?  bench.py alternates 2^20-pair
pricing calls over two streams of EQUAL priority; the dispatcher may then interleave the workgroups of both launches,
so both drain at the same time.  With one high- and one normal-priority stream the high one should always be served
first and the other fill what it leaves (its drain, the gap before its next launch).  200 steps each, best of 3, for
several stream pairs per class (which hardware queue a stream lands on matters, tools/queue_map_probe.py).
    python tools/priority_probe.py [log2 pairs = 20]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import hwmc  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
p = hwmc.default_params("head")
P, f = hwmc.analytic_curve(p)
engines = []
for _ in range(3):
    e = hwmc.HullWhiteEngine(p, device=0)
    e.load_market_data(P, f)
    engines.append(e)
K = 200
out = torch.zeros((K, 5), dtype=torch.float64, device=dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range reported by torch:", (lo, hi))
normal = [torch.cuda.Stream(device=dev, priority=0) for _ in range(4)]
high = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(4)]


def run(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        k = i % len(streams)
        engines[k].enqueue_ZBC(out[i].data_ptr(), n, 1234 + i, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    run(normal[:2])
print("%-34s %10s %12s" % ("streams", "ms/step", "M paths/s"))
cases = []
for i in range(4):
    for j in range(i + 1, 4):
        cases.append(("normal%d + normal%d" % (i, j), [normal[i], normal[j]]))
for i in range(4):
    for j in range(4):
        cases.append(("high%d + normal%d" % (i, j), [high[i], normal[j]]))
for i in range(4):
    for j in range(i + 1, 4):
        cases.append(("high%d + high%d" % (i, j), [high[i], high[j]]))
cases.append(("high0 + normal0 + normal1", [high[0], normal[0], normal[1]]))
cases.append(("normal0 alone", [normal[0]]))
cases.append(("high0 alone", [high[0]]))
best = {}
for name, streams in cases:
    t = min(run(streams) for _ in range(3))
    cls = " + ".join(s.rstrip("0123456789") for s in name.split(" + "))
    best[cls] = min(best.get(cls, 1e9), t)
    print("%-34s %10.4f %12.0f" % (name, t, 2 * n / t / 1e3), flush=True)
print("best per class:")
for cls, t in best.items():
    print("  %-30s %10.4f %12.0f" % (cls, t, 2 * n / t / 1e3))
