#!/usr/bin/env python3
"""What does the per-pricing-call all-reduce cost a pipelined run on ONE GPU (one-rank RCCL communicator)?  200 steps of
2^20 pairs alternating over two contexts/streams: (a) nothing else, (b) + an event of the step's stream that a side stream
waits for, (c) + the 5-double all-reduce on that side stream (what bench.py --gpus N does per step), (d) the all-reduce
issued on the compute stream itself.    python tools/allreduce_cost_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
import hwmc  # noqa: E402

p = hwmc.default_params("head")
P, f = hwmc.analytic_curve(p)
engines = []
for _ in range(2):
    e = hwmc.HullWhiteEngine(p, device=0)
    e.load_market_data(P, f)
    engines.append(e)
K, n = 200, 1 << 20
out = torch.zeros((K, 5), dtype=torch.float64, device=dev)
warm = torch.zeros(5, dtype=torch.float64, device=dev)
dist.all_reduce(warm)
torch.cuda.synchronize()
cands = [torch.cuda.Stream(device=dev) for _ in range(8)]
side = torch.cuda.Stream(device=dev)


def run(streams, mode):
    works = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        k = i % 2
        engines[k].enqueue_ZBC(out[i].data_ptr(), n, 1234 + i, stream=streams[k].cuda_stream)
        if mode in ("event", "allreduce"):
            side.wait_stream(streams[k])
        if mode == "allreduce":
            with torch.cuda.stream(side):
                works.append(dist.all_reduce(out[i], async_op=True))
        if mode == "allreduce_same_stream":
            with torch.cuda.stream(streams[k]):
                works.append(dist.all_reduce(out[i], async_op=True))
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


for _ in range(3):
    run(cands[:2], "plain")
print("%-10s %-24s %10s %12s" % ("streams", "mode", "ms/step", "M paths/s"))
for pair in ((0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (2, 3)):
    streams = [cands[pair[0]], cands[pair[1]]]
    for mode in ("plain", "event", "allreduce", "allreduce_same_stream"):
        t = min(run(streams, mode) for _ in range(3))
        print("%-10s %-24s %10.4f %12.0f" % (pair, mode, t, 2 * n / t / 1e3), flush=True)
dist.destroy_process_group()
