"""CPU tests of the N>1 path: world_size-2 (and 3) gloo process groups exercise the sharding and
the all-reduce of the moment vector.  The compute on each rank is the CPU oracle here (the HIP
engine needs a GPU); on the GPU box bench.py runs the same partition with the engine and RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hw_oracle
    import hwmc
    p = hwmc.default_params("head")
    P, f = hwmc.analytic_curve(p)
    o = hw_oracle.Oracle("head")
    zbc = hwmc.parallel.sharded_sums(lambda cnt, off: o.zbc_moments(P, f, cnt, 77, path_offset=off, n_threads=1), n_total)
    curve = hwmc.parallel.sharded_sums(lambda cnt, off: o.zcb_curve(cnt, 5, path_offset=off, n_threads=1)[1:], 256)
    vega = hwmc.parallel.sharded_sums(lambda cnt, off: [o.vega_pathwise(P, f, cnt, 9, path_offset=off, n_threads=1)], n_total)
    price = hwmc.cv_price(zbc.numpy(), 2 * n_total, float(P[-1]))["price"]       # identical post-processing on every rank
    q.put((rank, zbc.tolist(), curve.tolist(), vega.tolist(), price, hwmc.parallel.shard(n_total, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_all_reduce_equals_single_process(world, oracle_mod, hwmc_mod):
    n_total = 1001                                   # not divisible by the world size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("head")
    whole = o.zbc_moments(P, f, n_total, 77)
    whole_curve = o.zcb_curve(256, 5)[1:]
    whole_vega = o.vega_pathwise(P, f, n_total, 9)
    shards = sorted(r[5] for r in results)
    assert shards[0][0] == 0 and sum(c for _, c in shards) == n_total            # exact integer path counts
    for (a, ca), (b, _) in zip(shards, shards[1:]):
        assert a + ca == b                                                       # contiguous, no overlap
    for _, zbc, curve, vega, price, _ in results:
        np.testing.assert_allclose(zbc, whole, rtol=1e-13)
        np.testing.assert_allclose(curve, whole_curve, rtol=1e-13)
        np.testing.assert_allclose(vega, [whole_vega], rtol=1e-12)
        assert price == pytest.approx(results[0][4], rel=0, abs=0)               # bit-identical across ranks


def test_shard_partition_properties(hwmc_mod):
    for n in (0, 1, 7, 1 << 20, (1 << 24) + 5):
        for world in (1, 2, 4, 8):
            parts = [hwmc_mod.parallel.shard(n, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == n
            assert parts[0][0] == 0
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def _strong_worker(rank, world, port, every, q):
    """The control flow bench.py --gpus N runs (strong scaling of ONE pricing workload, bucketed all-reduce of the
    per-step sums through hwmc.parallel.StepReducer), with the CPU oracle as the compute and gloo as the transport."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hw_oracle
    import hwmc
    K = 5
    n_total = hwmc.parallel.pricing_workload(world, "strong", pairs_log2=6, total_log2=9)      # 2^9 pairs per call whatever N > 1
    off, cnt = hwmc.parallel.shard(n_total, rank, world)
    p = hwmc.default_params("head")
    P, f = hwmc.analytic_curve(p)
    o = hw_oracle.Oracle("head")
    out = torch.zeros((K, 5), dtype=torch.float64)
    messages = []

    def reduce_rows(i0, i1, producers):
        messages.append((i0, i1, list(producers)))
        return dist.all_reduce(out[i0:i1 + 1], async_op=True)

    red = hwmc.parallel.StepReducer(every, K, reduce_rows)
    works = []
    for i in range(K):
        out[i] = torch.from_numpy(o.zbc_moments(P, f, cnt, 1234 + 12345 * i, path_offset=off, n_threads=1))
        works.append(red.after_step(i, producer=i % 2, last=(i == K - 1)))
    for w in works:
        if w is not None:
            w.wait()
    q.put((rank, n_total, (off, cnt), out.tolist(), messages))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("every,expected", [(1, [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4)]), (2, [(0, 1), (2, 3), (4, 4)]), (0, [(0, 4)])])
def test_strong_scaling_per_call_all_reduce(every, expected, oracle_mod, hwmc_mod):
    """N = 2, strong scaling: the two ranks share ONE 2^9-pair call per step; with every = 1 each pricing call is one
    collective (the default of bench.py), every = 2 buckets two steps, every = 0 sends one message at the end.  In
    every case each rank ends with the sums a single process computes for the whole call, to 1e-13."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, every, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("head")
    assert hwmc_mod.parallel.pricing_workload(1, "strong", 6, 9) == 64             # N = 1 keeps configs[1]'s size
    assert hwmc_mod.parallel.pricing_workload(4, "weak", 6, 9) == 4 * 64
    for rank, n_total, (off, cnt), rows, messages in results:
        assert n_total == 512 and cnt == 256 and off == 256 * rank
        assert [(a, b) for a, b, _ in messages] == expected
        assert all(set(prods) <= {0, 1} and prods for _, _, prods in messages)   # the streams whose launches the message waits for
        for i, row in enumerate(rows):
            np.testing.assert_allclose(row, o.zbc_moments(P, f, 512, 1234 + 12345 * i), rtol=1e-13)
