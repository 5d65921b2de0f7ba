#!/usr/bin/env python3
"""Which pairs of HIP streams actually run kernels concurrently?  HIP multiplexes streams onto a few hardware
queues; two streams on the same queue serialise.  Two contexts alternate 2^20-pair launches over every pair of
N streams; ms/step well below the serial figure means the pair overlaps.

    python tools/queue_map_probe.py [--n 8] [--kind torch|hip]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import hwmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--kind", default="torch")
    a = ap.parse_args()
    n = 1 << 20
    dev = torch.device("cuda", 0)
    p = hwmc.default_params("head")
    eng = []
    for _ in range(2):
        e = hwmc.HullWhiteEngine(p, device=0)
        e.load_market_data(*hwmc.analytic_curve(p))
        eng.append(e)
    if a.kind == "torch":
        keep = [torch.cuda.Stream(device=dev) for _ in range(a.n)]
        handles = [s.cuda_stream for s in keep]
    elif a.kind == "torch-prio":                       # first half normal priority, second half high priority
        keep = [torch.cuda.Stream(device=dev, priority=0 if k < a.n // 2 else -1) for k in range(a.n)]
        handles = [s.cuda_stream for s in keep]
        print("priorities:", [s.priority for s in keep])
    else:
        hip = ctypes.CDLL("libamdhip64.so")
        handles = []
        for _ in range(a.n):
            h = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
            handles.append(h.value)
    out = torch.zeros((a.steps, 5), dtype=torch.float64, device=dev)

    def run(i, j):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            eng[k % 2].enqueue_ZBC(out[k].data_ptr(), n, 1 + k, stream=handles[i if k % 2 == 0 else j])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / a.steps

    t_end = time.time() + 0.5
    while time.time() < t_end:
        run(0, 0)
    print("ms/step, row = stream of even steps, column = stream of odd steps (diagonal = serial)")
    for i in range(a.n):
        row = []
        for j in range(a.n):
            row.append(min(run(i, j) for _ in range(3)))
        print(" ".join(f"{t:.4f}" for t in row))


if __name__ == "__main__":
    main()
