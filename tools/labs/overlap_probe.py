#!/usr/bin/env python3
"""Does alternating independent pricing batches over S contexts (S streams) on ONE GPU hide the ~15 us fixed
cost (launch + fill/drain) each 2^20-pair launch pays when the batches run strictly one after another?

    python tools/overlap_probe.py [--log2 20] [--steps 40] [--streams 1 2 3 4]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import hwmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    n = 1 << a.log2
    dev = torch.device("cuda", 0)
    p = hwmc.default_params("head")
    S = max(a.streams)
    eng, streams = [], []
    for _ in range(S):
        e = hwmc.HullWhiteEngine(p, device=0)
        e.load_market_data(*hwmc.analytic_curve(p))
        eng.append(e)
        streams.append(torch.cuda.Stream(device=dev))
    out = torch.zeros((a.steps, 5), dtype=torch.float64, device=dev)
    ref = None

    def run(s):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            eng[i % s].enqueue_ZBC(out[i].data_ptr(), n, 1234 + 12345 * i, stream=streams[i % s].cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / a.steps

    t_end = time.time() + 0.5
    while time.time() < t_end:
        run(1)
    res = {s: [] for s in a.streams}
    for _ in range(a.rounds):
        for s in a.streams:
            res[s].append(run(s))
            got = out.cpu().numpy().copy()
            if ref is None:
                ref = got
            assert np.array_equal(ref, got), "results depend on the stream count"
    print(f"2^{a.log2} pairs per step, {a.steps} steps per run, wall clock incl. launch")
    for s in a.streams:
        t = np.array(res[s])
        print(f"streams {s}: med {np.median(t):.4f} ms/step  min {t.min():.4f}  -> {2 * n / np.median(t) / 1e3:.1f} M paths/s")


if __name__ == "__main__":
    main()
