"""Lifecycle and failure-injection cases of the C ABI, each meant to run in a FRESH process:

    python tests/hazard_cases.py <case>

tests/test_gpu_hazards.py starts one child per case and asserts exit code 0, so a crash or a hang in one of them is one
failed test with its output attached -- it cannot take the parity suite's process (and pass count) with it, which is
what happened in round 2 (GPUTEST_r02.json: SIGSEGV inside the old stream guard).  Prints "OK <case>" on success.

Rules under test (include/hw_mc.h):
  * stream lifetime: a stream given to an *_async call only has to live for the duration of that call;
  * hw_comm_*: a rank that fails before the collective makes the call return its error with nobody waiting in the
    all-reduce and the communicator stays usable; a failure inside the collective phase aborts the communicator.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _hip():
    """The HIP runtime THIS process already runs on (torch's bundled copy or the system's, whichever was loaded first)."""
    loaded = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
    assert loaded, "no HIP runtime mapped"
    return ctypes.CDLL(loaded[0])


def _setup(profile="head"):
    import torch
    import hwmc
    hwmc.build_library()
    torch.zeros(1, device="cuda")
    p = hwmc.default_params(profile)
    P, f = hwmc.analytic_curve(p)
    return torch, hwmc, p, P, f


def _new_stream(hip):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0          # hipStreamNonBlocking
    return s


def case_destroyed_stream_then_another_stream():
    """Round 2's crash, verbatim: enqueue on s1, synchronise and destroy s1, enqueue on s2."""
    torch, hwmc, p, P, f = _setup()
    hip = _hip()
    dev = torch.device("cuda", 0)
    out = torch.zeros((2, 5), dtype=torch.float64, device=dev)
    with hwmc.HullWhiteEngine(p, device=0) as e:
        e.load_market_data(P, f)
        s1 = _new_stream(hip)
        e.enqueue_ZBC(out[0].data_ptr(), 5000, 3, stream=s1.value)
        assert hip.hipStreamSynchronize(s1) == 0
        assert hip.hipStreamDestroy(s1) == 0
        s2 = torch.cuda.Stream(device=dev)
        e.enqueue_ZBC(out[1].data_ptr(), 5000, 4, stream=s2.cuda_stream)            # previous stream is gone
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[0], e.simulate_ZBC_control_variate(5000, 3))
        assert np.array_equal(got[1], e.simulate_ZBC_control_variate(5000, 4))


def case_stream_churn():
    """200 short-lived caller streams through ONE context, each destroyed right after its call WITHOUT a synchronisation
    (hipStreamDestroy lets the work finish), mixed with calls on the context's own stream and synchronous calls: every
    result bit-equal to the synchronous call, i.e. the launches were serialised through the context's event, never
    through a dead handle.  Freed stream objects are recycled by the runtime's allocator here, which is what made the old
    guard's use-after-free visible."""
    torch, hwmc, p, P, f = _setup()
    hip = _hip()
    dev = torch.device("cuda", 0)
    K = 200
    out = torch.zeros((K, 5), dtype=torch.float64, device=dev)
    with hwmc.HullWhiteEngine(p, device=0) as e:
        e.load_market_data(P, f)
        want = np.stack([e.simulate_ZBC_control_variate(3000 + 7 * i, 50 + i) for i in range(K)])
        for i in range(K):
            if i % 5 == 4:
                e.enqueue_ZBC(out[i].data_ptr(), 3000 + 7 * i, 50 + i)                # the context's own stream
                continue
            s = _new_stream(hip)
            e.enqueue_ZBC(out[i].data_ptr(), 3000 + 7 * i, 50 + i, stream=s.value)
            assert hip.hipStreamDestroy(s) == 0                                       # no synchronisation first
            if i % 17 == 0:
                assert np.array_equal(e.simulate_ZBC_control_variate(999, 1), e.simulate_ZBC_control_variate(999, 1))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)


def case_destroy_context_after_its_last_stream_is_gone():
    """hw_destroy with the context's latest work on a caller's stream that no longer exists; and the other order: the
    context destroyed first, the stream afterwards."""
    torch, hwmc, p, P, f = _setup()
    hip = _hip()
    dev = torch.device("cuda", 0)
    out = torch.zeros((4, 5), dtype=torch.float64, device=dev)
    for order in ("stream_first", "context_first", "no_sync"):
        e = hwmc.HullWhiteEngine(p, device=0)
        e.load_market_data(P, f)
        s = _new_stream(hip)
        e.enqueue_ZBC(out[0].data_ptr(), 1 << 18, 3, stream=s.value)
        e.enqueue_zcb(torch.zeros(p.n_mat, dtype=torch.float64, device=dev).data_ptr(), 4096, 5, stream=s.value)
        if order == "stream_first":
            assert hip.hipStreamSynchronize(s) == 0 and hip.hipStreamDestroy(s) == 0
            e.close()
        elif order == "context_first":
            e.close()                                                                 # waits on the context's own event
            assert hip.hipStreamDestroy(s) == 0
        else:
            assert hip.hipStreamDestroy(s) == 0                                       # work still in flight
            e.close()
    torch.cuda.synchronize()
    with hwmc.HullWhiteEngine(p, device=0) as e:
        e.load_market_data(P, f)
        assert np.array_equal(out[0].cpu().numpy(), e.simulate_ZBC_control_variate(1 << 18, 3))


def _fleet(hwmc, p, P, f, G, host_sum):
    engines = [hwmc.HullWhiteEngine(p, device=0) for _ in range(G)]
    for e in engines:
        e.load_market_data(P, f)
    return engines, hwmc.HullWhiteFleet(engines, host_sum=host_sum)


def case_fleet_rank_fails_before_the_collective():
    """Four worker threads on one GPU (HW_COMM_HOST_SUM), rank 2 fails in phase 1 of its first two jobs: each call comes
    back with rank 2's text in bounded time -- nobody waits for the missing rank -- and the third call is correct."""
    os.environ["HWMC_COMM_INJECT_FAIL"] = "2:1:2"
    torch, hwmc, p, P, f = _setup("readme")
    engines, fleet = _fleet(hwmc, p, P, f, 4, True)
    try:
        for _ in range(2):
            t0 = time.perf_counter()
            try:
                fleet.simulate_ZBC_control_variate(1 << 18, 77)
                raise AssertionError("the injected failure did not surface")
            except hwmc.HwError as ex:
                assert "injected failure on rank 2 before the collective" in str(ex) and "device 0" in str(ex), str(ex)
            assert time.perf_counter() - t0 < 5.0
        got = fleet.simulate_ZBC_control_variate(50_001, 77)
        want = np.zeros(5)
        for g in range(4):
            off, cnt = hwmc.parallel.shard(50_001, g, 4)
            want = want + engines[0].simulate_ZBC_control_variate(cnt, 77, path_offset=off)
        assert np.array_equal(got, want)
        c = fleet.simulate_zcb(20_000, 5)
        assert c[0] == 2.0 * 20_000
    finally:
        fleet.close()
        for e in engines:
            e.close()


def case_fleet_rccl_rank_fails_before_the_collective():
    """The same through the RCCL path (one rank: all a one-GPU box allows): the call returns the error, no all-reduce is
    enqueued, the communicator stays usable and the next call equals the single-context call bit for bit."""
    os.environ["HWMC_COMM_INJECT_FAIL"] = "0:1:1"
    torch, hwmc, p, P, f = _setup("readme")
    engines, fleet = _fleet(hwmc, p, P, f, 1, False)
    try:
        try:
            fleet.simulate_ZBC_control_variate(1 << 16, 77)
            raise AssertionError("the injected failure did not surface")
        except hwmc.HwError as ex:
            assert "injected failure on rank 0 before the collective" in str(ex), str(ex)
        assert np.array_equal(fleet.simulate_ZBC_control_variate(50_001, 77), engines[0].simulate_ZBC_control_variate(50_001, 77))
    finally:
        fleet.close()
        for e in engines:
            e.close()


def case_fleet_rccl_rank_fails_inside_the_collective_phase():
    """A failure AFTER the rendezvous cannot be undone rank by rank: the failing worker aborts the communicators
    (ncclCommAbort releases ranks blocked in the all-reduce), the call returns its error, later calls are refused, and
    hw_comm_destroy returns."""
    os.environ["HWMC_COMM_INJECT_FAIL"] = "0:2:1"
    torch, hwmc, p, P, f = _setup("readme")
    engines, fleet = _fleet(hwmc, p, P, f, 1, False)
    try:
        t0 = time.perf_counter()
        try:
            fleet.simulate_ZBC_control_variate(1 << 16, 77)
            raise AssertionError("the injected failure did not surface")
        except hwmc.HwError as ex:
            assert "inside the collective phase" in str(ex), str(ex)
        try:
            fleet.simulate_ZBC_control_variate(1 << 16, 77)
            raise AssertionError("an aborted communicator accepted a call")
        except hwmc.HwError as ex:
            assert "aborted" in str(ex), str(ex)
        assert time.perf_counter() - t0 < 10.0
    finally:
        fleet.close()
    # the contexts are still good on their own
    assert engines[0].simulate_ZBC_control_variate(1000, 1)[0] > 0
    for e in engines:
        e.close()


def case_fleet_host_sum_rank_fails_after_the_rendezvous():
    """HW_COMM_HOST_SUM has no collective to abort: a phase-2 failure on one rank is reported and the fleet goes on."""
    os.environ["HWMC_COMM_INJECT_FAIL"] = "3:2:1"
    torch, hwmc, p, P, f = _setup("readme")
    engines, fleet = _fleet(hwmc, p, P, f, 4, True)
    try:
        try:
            fleet.simulate_ZBC_control_variate(1 << 16, 77)
            raise AssertionError("the injected failure did not surface")
        except hwmc.HwError as ex:
            assert "rank 3 inside the collective phase" in str(ex), str(ex)
        a = fleet.simulate_ZBC_control_variate(4096, 1)
        assert np.array_equal(a, fleet.simulate_ZBC_control_variate(4096, 1)) and a[0] > 0
    finally:
        fleet.close()
        for e in engines:
            e.close()


CASES = {k[5:]: v for k, v in list(globals().items()) if k.startswith("case_")}

if __name__ == "__main__":
    name = sys.argv[1]
    CASES[name]()
    print("OK", name)
