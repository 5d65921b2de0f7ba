#!/usr/bin/env python3
"""How do two queues settle into their rhythm after a synchronisation?  40 pricing calls of 2^20 pairs alternating over two
contexts/streams, a HIP event pair around EVERY path-kernel launch (hw_profile_enable: costs queue bubbles, the picture is
qualitative): start, end and duration of each launch on a common clock.
    python tools/ramp_probe.py [pattern = ab | aab | abc]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import hwmc  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n = 1 << 20
pattern = sys.argv[1] if len(sys.argv) > 1 else "ab"
p = hwmc.default_params("head")
P, f = hwmc.analytic_curve(p)
engines = []
for _ in range(3):
    e = hwmc.HullWhiteEngine(p, device=0)
    e.load_market_data(P, f)
    engines.append(e)
out = torch.zeros((512, 5), dtype=torch.float64, device=dev)
cands = [torch.cuda.Stream(device=dev) for _ in range(8)]
K = 40


def order(K):
    if pattern == "ab":
        return [i % 2 for i in range(K)]
    if pattern == "abc":
        return [i % 3 for i in range(K)]
    if pattern == "aab":                      # two calls on queue a first: queue b starts half a beat late?
        return [0] + [(i + 1) % 2 for i in range(K - 1)]
    raise SystemExit("pattern?")


def run(streams, K, ks, delay_us=0.0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        k = ks[i]
        engines[k].enqueue_ZBC(out[i].data_ptr(), n, 1234 + i, stream=streams[k].cuda_stream)
        if i == 0 and delay_us > 0.0:         # the host holds the second queue's first launch back: the queues start staggered
            t_end = t0 + delay_us * 1e-6
            while time.perf_counter() < t_end:
                pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    run(cands[:1] * 3, 16, [0] * 16)
best = (1e9, None)
for i in range(8):
    for j in range(i + 1, 8):
        t = min(run([cands[i], cands[j], cands[(j + 1) % 8]], 16, [q % 2 for q in range(16)]) for _ in range(2))
        if t < best[0]:
            best = (t, (i, j))
i, j = best[1]
third = [q for q in range(8) if q not in (i, j)]
streams = [cands[i], cands[j], cands[third[0]]]
ks = order(K)
for kk in (20, 40):
    t = np.median([run(streams, kk, order(kk)) for _ in range(9)])
    print("pattern %s, K = %d, no event pairs: %.4f ms per call" % (pattern, kk, t / kk))
if pattern == "ab":
    for d in (0, 20, 40, 60, 80, 100, 120, 150, 0):
        for kk in (20, 40):
            t = np.median([run(streams, kk, order(kk), delay_us=d) for _ in range(15)])
            print("second queue held back %3d us, K = %d: %.4f ms per call" % (d, kk, t / kk), flush=True)
for eng in engines:
    eng.profile_enable(K)
run(streams, K, ks)
spans = {}
for k, eng in enumerate(engines):
    st, en = eng.profile_spans(ref=engines[0])
    cnt = 0
    for i in range(K):
        if ks[i] == k:
            spans[i] = (float(st[cnt]), float(en[cnt]), k)
            cnt += 1
t0 = min(v[0] for v in spans.values())
print("call queue   start us     end us   length us   (event pairs around every launch)")
for i in range(K):
    s, e_, k = spans[i]
    print("%4d   %s   %9.1f  %9.1f  %9.1f" % (i, "abc"[k], (s - t0) * 1e3, (e_ - t0) * 1e3, (e_ - s) * 1e3))
ends = sorted(v[1] for v in spans.values())
print("completion intervals (us):", " ".join("%.0f" % ((b - a) * 1e3) for a, b in zip(ends[:-1], ends[1:])))
