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
