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


def _selfcheck_worker(rank, world, port, wrong_seed_rank, q):
    """Two ranks build what bench.py --gpus 2 puts into its line's self-check fields (hwmc.parallel.world_size_seen,
    launch_skew, gather_floats) around an oracle-computed, gloo-reduced pricing call; rank 0 hands the line out."""
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hw_oracle
    import hwmc
    par = hwmc.parallel
    n_total = par.pricing_workload(world, "strong", pairs_log2=6, total_log2=9)
    off, cnt = par.shard(n_total, rank, world)
    p = hwmc.default_params("head")
    P, f = hwmc.analytic_curve(p)
    o = hw_oracle.Oracle("head")
    starts = []
    for _ in range(3):                                     # three "timed regions": barrier, note the host clock, work
        dist.barrier()
        starts.append(time.time())
        time.sleep(0.001 * rank)
    seed = 1234 + (1 if rank == wrong_seed_rank else 0)    # a rank pricing another sample set must be caught
    t0 = time.perf_counter()
    sums = torch.from_numpy(o.zbc_moments(P, f, cnt, seed, path_offset=off, n_threads=1)).clone()
    kernel_ms = (time.perf_counter() - t0) * 1e3
    par.allreduce_sums(sums)
    step_ms = (time.perf_counter() - t0) * 1e3
    tmax = torch.tensor([step_ms], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    mg = {"world_size_seen": par.world_size_seen(), "launch_skew": par.launch_skew(starts),
          "per_rank_kernel_ms": [r[0] for r in par.gather_floats([kernel_ms])],
          "compute_only_ms_per_step": None, "exposed_allreduce_ms_per_step": 0.0, "single_call_ms": step_ms, "single_call_host_sum_ms": step_ms}
    ms = float(tmax.item())
    line = {"metric": "M paths/sec (ZBC option, 1000 steps)", "unit": "M paths/s", "dtype": "f32", "n_gpus": world, "scaling": "strong",
            "ms_per_step": ms, "value": 2 * n_total / ms / 1000.0, "sums_step0": sums.tolist(), "multi_gpu": mg, "rehearsal": "cpu oracle over gloo",
            "config": {"grid_steps": 1000, "steps_executed": 500, "stepper": "superposed", "theta_profile": "head", "pairs_per_step": n_total,
                       "effective_paths": 2 * n_total, "streams_per_gpu": 1}}
    if rank == 0:
        q.put(line)
    dist.barrier()
    dist.destroy_process_group()


def _two_rank_line(wrong_seed_rank=-1):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_selfcheck_worker, args=(r, world, port, wrong_seed_rank, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    line = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return line


def test_an_n_gpu_line_checks_itself_against_the_n1_line(oracle_mod, hwmc_mod):
    """no N > 1 run has ever executed on hardware, so the first one must be self-checking.
    hwmc.parallel.check_scaling_lines takes the N = 1 line (with strong_scaling_reference.sums_step0) and an N-GPU line and
    asserts: every rank took part (torch's world size == an all-reduce of ones == n_gpus), the SAME sample set was priced
    (sums of the first timed step equal to 1e-13), value == paths / time; its report carries the fields that decide between
    '>= 7x' and 'collective exposed'.  Here the two ranks' line is built over gloo with the CPU oracle as the compute."""
    import copy
    import time
    par = hwmc_mod.parallel
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    o = oracle_mod.Oracle("head")
    t0 = time.perf_counter()
    whole = o.zbc_moments(P, f, 512, 1234, n_threads=1)
    ms1 = (time.perf_counter() - t0) * 1e3
    line1 = {"metric": "M paths/sec (ZBC option, 1000 steps)", "unit": "M paths/s", "dtype": "f32", "n_gpus": 1, "scaling": "strong",
             "config": {"grid_steps": 1000, "steps_executed": 500, "stepper": "superposed", "theta_profile": "head", "pairs_per_step": 64},
             "strong_scaling_reference": {"pairs_per_step": 512, "n_gpus": 1, "ms_per_step": ms1, "serial_ms_per_step": ms1, "kernel_ms": ms1,
                                          "sums_step0": [float(x) for x in whole]}}
    lineN = _two_rank_line()
    rep = par.check_scaling_lines(line1, lineN)
    assert rep["n_gpus"] == 2 and rep["sums_step0_max_rel_diff"] <= 1e-13
    assert lineN["multi_gpu"]["world_size_seen"] == {"dist": 2, "allreduce_of_ones": 2}
    sk = lineN["multi_gpu"]["launch_skew"]
    assert sk["regions"] == 3 and 0.0 <= sk["median_ms"] <= sk["max_ms"] < 1000.0
    assert len(lineN["multi_gpu"]["per_rank_kernel_ms"]) == 2 and rep["strong_scaling"] > 0 and "verdict" in rep
    # a rank that priced another sample set (wrong seed on rank 1) is caught by the sums, nothing else would notice
    with pytest.raises(AssertionError, match="did not price the N = 1 sample set"):
        par.check_scaling_lines(line1, _two_rank_line(wrong_seed_rank=1))
    # a rank missing from the collective / a launcher that started fewer processes
    bad = copy.deepcopy(lineN)
    bad["multi_gpu"]["world_size_seen"]["allreduce_of_ones"] = 1
    with pytest.raises(AssertionError, match="ranks seen"):
        par.check_scaling_lines(line1, bad)
    bad = copy.deepcopy(lineN)
    bad["config"]["pairs_per_step"] = 256
    with pytest.raises(AssertionError, match="pairs per call"):
        par.check_scaling_lines(line1, bad)
    bad = copy.deepcopy(lineN)
    bad["value"] *= 1.5
    with pytest.raises(AssertionError, match="effective_paths / ms_per_step"):
        par.check_scaling_lines(line1, bad)
    with pytest.raises(AssertionError, match="no strong_scaling_reference"):
        par.check_scaling_lines({k: v for k, v in line1.items() if k != "strong_scaling_reference"}, lineN)
    # the three verdicts, on synthetic timings of an 8-GPU line: scales / compute-bound shortfall / collective exposed
    def synth(step_ms, kernel_ms, exposed):
        d = copy.deepcopy(lineN)
        d["n_gpus"] = 8
        d["rehearsal"] = "synthetic"
        d["ms_per_step"] = step_ms
        d["value"] = d["config"]["effective_paths"] / step_ms / 1000.0
        d["multi_gpu"].update(world_size_seen={"dist": 8, "allreduce_of_ones": 8}, per_rank_kernel_ms=[kernel_ms] * 8,
                              exposed_allreduce_ms_per_step=exposed, launch_skew={"median_ms": 0.004, "max_ms": 0.01, "regions": 7})
        return d
    ref = copy.deepcopy(line1)
    ref["strong_scaling_reference"].update(ms_per_step=2.74, serial_ms_per_step=2.77, kernel_ms=2.77)
    assert par.check_scaling_lines(ref, synth(0.36, 0.355, 0.004))["verdict"].startswith("scales: 7.61x on 8 GPUs")
    assert par.check_scaling_lines(ref, synth(0.45, 0.44, 0.005))["verdict"].startswith("compute-bound shortfall")
    assert par.check_scaling_lines(ref, synth(0.45, 0.355, 0.09))["verdict"].startswith("collective exposed")


def test_an_n_gpu_line_proves_itself_against_committed_constants(oracle_mod, hwmc_mod, monkeypatch):
    """the first N > 1 run on hardware will arrive WITHOUT a companion N = 1 line from the same box, so the
    one-GPU side of the comparison is committed (reference_sums.json beside hwmc.parallel: whole-call sums per configuration
    -- deterministic, the driver's box reproduced the builder's to the last bit --, the one-GPU time, the predicted N-GPU
    times) and bench.py --gpus N puts hwmc.parallel.self_check(line) into multi_gpu.self_check, exiting non-zero when the
    ranks seen are not --gpus or the sums differ by more than 1e-13.  Here a two-rank line built over gloo (CPU oracle as
    the compute) goes through it against a store holding the oracle's own whole-call sums."""
    import copy
    par = hwmc_mod.parallel
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    whole = [float(x) for x in oracle_mod.Oracle("head").zbc_moments(P, f, 512, 1234, n_threads=1)]
    key = par.reference_key("zbc", "head", "f32", "superposed", 1000, 9, 1234)
    store = {key: whole,
             "_one_gpu_ms": {key.rsplit("|", 1)[0]: {"ms_per_step": 40.0, "serial_ms_per_step": 41.0, "kernel_ms": 40.0, "source": "test"}},
             "_predicted_ms_per_step": {key.rsplit("|", 1)[0]: {"2": {"throughput_mode": 20.5, "latency_mode": 21.0}}}}
    monkeypatch.setattr(par, "_reference_cache", store)
    lineN = _two_rank_line()
    sc = par.self_check(lineN, 2)
    assert sc["failed"] == [] and sc["reference"] == key and sc["sums_max_rel_diff"] <= 1e-13
    assert sc["world_size_seen"] == {"dist": 2, "allreduce_of_ones": 2} and sc["gpus_requested"] == 2
    assert sc["one_gpu_ms_per_step"] == 40.0 and sc["strong_scaling"] == pytest.approx(40.0 / lineN["ms_per_step"])
    assert sc["predicted_ms_per_step"]["throughput_mode"] == 20.5 and sc["ms_per_step_over_prediction"] == pytest.approx(lineN["ms_per_step"] / 20.5)
    assert sc["verdict"].split(":")[0] in ("scales", "compute-bound shortfall", "collective exposed")
    # a rank that priced another sample set: caught by the committed sums alone, and named as the reason to exit non-zero
    bad = par.self_check(_two_rank_line(wrong_seed_rank=1), 2)
    assert len(bad["failed"]) == 1 and "another sample set" in bad["failed"][0] and bad["verdict"].startswith("FAILED")
    # fewer ranks than --gpus (a launcher that started 2 of 8), a rank missing from the collective
    short = par.self_check(lineN, 8)
    assert any("--gpus 8" in x for x in short["failed"])
    miss = copy.deepcopy(lineN)
    miss["multi_gpu"]["world_size_seen"]["allreduce_of_ones"] = 1
    assert any("ranks seen" in x for x in par.self_check(miss, 2)["failed"])
    # a configuration without a committed constant is reported, not failed (the 2^18-pair rehearsal, weak scaling)
    other = copy.deepcopy(lineN)
    other["config"]["grid_steps"] = 4000
    sc = par.self_check(other, 2)
    assert sc["failed"] == [] and sc["sums_max_rel_diff"] is None and "no committed one-GPU constant" in sc["verdict"]
    # the committed store itself: BASELINE configs[1] and configs[3] are there, with the 2^24-pair call's time and predictions
    monkeypatch.setattr(par, "_reference_cache", None)
    real = par.load_reference_sums()
    k24 = par.reference_key("zbc", "head", "f32", "superposed", 1000, 24, 1234)
    assert len(real[k24]) == 5 and len(real[par.reference_key("zbc", "head", "f32", "superposed", 1000, 20, 1234)]) == 5
    assert real["_one_gpu_ms"][k24.rsplit("|", 1)[0]]["ms_per_step"] > 2.0
    pred = real["_predicted_ms_per_step"][k24.rsplit("|", 1)[0]]
    assert set(pred) >= {"2", "4", "8"} and pred["8"]["throughput_mode"] < pred["4"]["throughput_mode"] < pred["2"]["throughput_mode"]
    assert real["_one_gpu_ms"][k24.rsplit("|", 1)[0]]["ms_per_step"] / pred["8"]["throughput_mode"] >= 7.0     # the north star's >= 7x, as predicted
