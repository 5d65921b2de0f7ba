"""GPU tests of the stream rules of the C ABI (include/hw_mc.h, "Streams and the context's workspaces"): a context's
workspaces are shared by all its launches, so calls on different streams through ONE context must serialise on the
device (not race), the workspace must be able to grow while launches are in flight, and the pinned staging ring of
the small table uploads must survive more uploads in flight than it has slots."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(hwmc_mod):
    import torch
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    e = hwmc_mod.HullWhiteEngine(p, device=0)
    e.load_market_data(P, f)
    yield hwmc_mod, torch, e
    e.close()


def test_one_context_shared_by_several_streams_does_not_race(ctx):
    hwmc, torch, e = ctx
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    n, K = 1 << 18, 24
    want = np.stack([e.simulate_ZBC_control_variate(n, 100 + i) for i in range(K)])
    out = torch.zeros((K, 5), dtype=torch.float64, device=dev)
    for rep in range(3):
        out.zero_()
        torch.cuda.synchronize()
        for i in range(K):                                            # back to back, no host synchronisation, 4 streams, ONE context
            e.enqueue_ZBC(out[i].data_ptr(), n, 100 + i, stream=streams[(i * 7 + rep) % 4].cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), want)       # bit-identical: same launches, serialised by the guard
    # mixed kernels through the same context and streams: curve (uploads a scale table), Vega, reference stepper (uploads a drift table)
    curve = torch.zeros((6, e.params.n_mat), dtype=torch.float64, device=dev)
    vega = torch.zeros((6, 1), dtype=torch.float64, device=dev)
    ref = torch.zeros((6, 5), dtype=torch.float64, device=dev)
    sig = [0.1, 0.101, 0.099, 0.1, 0.102, 0.098]
    for i in range(6):
        e.enqueue_zcb(curve[i].data_ptr(), 4096, 5, sigma=sig[i], stream=streams[i % 4].cuda_stream)
        e.enqueue_sensitivity(vega[i].data_ptr(), 1 << 14, 9 + i, stream=streams[(i + 1) % 4].cuda_stream)
        e.enqueue_ZBC(ref[i].data_ptr(), 1 << 14, 3, sigma=sig[i], drift_mode=hwmc.HW_DRIFT_SHIFTED,
                      stepper=hwmc.HW_STEPPER_REFERENCE, stream=streams[(i + 2) % 4].cuda_stream)
    torch.cuda.synchronize()
    for i in range(6):
        np.testing.assert_array_equal(curve[i].cpu().numpy(), e.simulate_zcb(4096, 5, sigma=sig[i]))
        assert float(vega[i, 0]) == e.simulate_sensitivity(1 << 14, 9 + i)
        np.testing.assert_array_equal(ref[i].cpu().numpy(), e.simulate_ZBC_control_variate(
            1 << 14, 3, sigma=sig[i], drift_mode=hwmc.HW_DRIFT_SHIFTED, stepper=hwmc.HW_STEPPER_REFERENCE))


def test_workspace_grows_under_launches_in_flight(hwmc_mod):
    import torch
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sizes = [1 << 12, 1 << 16, 1 << 20, 1 << 14, 3 << 20, 1 << 10, 1 << 22]         # every growth happens behind unfinished launches
    with hwmc_mod.HullWhiteEngine(p, device=0) as e:
        e.load_market_data(P, f)
        out = torch.zeros((len(sizes), 5), dtype=torch.float64, device=dev)
        for i, n in enumerate(sizes):
            e.enqueue_ZBC(out[i].data_ptr(), n, 40 + i, stream=st.cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i, n in enumerate(sizes):
            np.testing.assert_array_equal(got[i], e.simulate_ZBC_control_variate(n, 40 + i))


def test_more_table_uploads_in_flight_than_staging_slots(hwmc_mod):
    import torch
    p = hwmc_mod.default_params("readme")
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sig = [0.09 + 0.002 * i for i in range(12)]                                       # 12 different scale tables, 4 slots
    with hwmc_mod.HullWhiteEngine(p, device=0) as e:
        out = torch.zeros((len(sig), p.n_mat), dtype=torch.float64, device=dev)
        for i, s in enumerate(sig):
            e.enqueue_zcb(out[i].data_ptr(), 20000, 77, sigma=s, stream=st.cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i, s in enumerate(sig):
            np.testing.assert_array_equal(got[i], e.simulate_zcb(20000, 77, sigma=s))
        assert len({tuple(r) for r in got}) == len(sig)                               # and they ARE different tables


# (what a caller may do with a stream AFTER a call -- destroy it, before or after the context -- is exercised by
# tests/test_gpu_hazards.py, one child process per case)


def test_polled_completion_equals_stream_synchronisation(hwmc_mod, monkeypatch):
    """A synchronous call that is not asked for the kernel time returns when the completion stamps of ITS results have
    landed in the pinned host buffer (include/hw_mc.h); HWMC_SPIN_WAIT=0 (read by hw_create) waits for the stream instead.
    Same kernels, so the sums must be bit-identical: small launches (the stamps arrive while the host polls), launches
    longer than the 300 us polling window (the fallback), every synchronous entry point, many calls in a row (a stale stamp
    of an earlier call must never be taken for the current one), and interleaved with calls that DO ask for the time."""
    p = hwmc_mod.default_params("readme")
    P, f = hwmc_mod.analytic_curve(p)
    polled = hwmc_mod.HullWhiteEngine(p, device=0)
    monkeypatch.setenv("HWMC_SPIN_WAIT", "0")
    waited = hwmc_mod.HullWhiteEngine(p, device=0)
    monkeypatch.delenv("HWMC_SPIN_WAIT")
    try:
        for e in (polled, waited):
            e.load_market_data(P, f)
        for n in (1, 777, 1 << 14, 1 << 20, 1 << 23):                        # 2^23 pairs: 1.4 ms, far beyond the polling window
            a, b = polled.simulate_ZBC_control_variate(n, 5), waited.simulate_ZBC_control_variate(n, 5)
            assert np.array_equal(a, b) and a[0] > 0, n
        for i in range(300):                                                 # back to back, alternating sizes: no stale stamps
            n = 4096 if i % 2 else 300
            assert np.array_equal(polled.simulate_ZBC_control_variate(n, 100 + i), waited.simulate_ZBC_control_variate(n, 100 + i)), i
        assert np.array_equal(polled.simulate_zcb(50_000, 3), waited.simulate_zcb(50_000, 3))
        assert polled.simulate_sensitivity(1 << 16, 9) == waited.simulate_sensitivity(1 << 16, 9)
        assert np.array_equal(polled.simulate_ZBC_runs(30_000, 1234, n_runs=20), waited.simulate_ZBC_runs(30_000, 1234, n_runs=20))
        assert np.array_equal(polled.simulate_ZBC_scenarios(1 << 15, 7, [0.099, 0.1, 0.101], [1, 0, 1]),
                              waited.simulate_ZBC_scenarios(1 << 15, 7, [0.099, 0.1, 0.101], [1, 0, 1]))
        assert np.array_equal(polled.simulate_sensitivity_runs(20_000, 4, n_runs=20), waited.simulate_sensitivity_runs(20_000, 4, n_runs=20))
        for i in range(20):                                                  # with and without the event pair, mixed
            s, ms = polled.simulate_ZBC_control_variate(9999, i, return_ms=True)
            assert ms > 0 and np.array_equal(s, polled.simulate_ZBC_control_variate(9999, i))
    finally:
        polled.close()
        waited.close()
