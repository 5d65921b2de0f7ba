"""hw_pipe_* (include/hw_mc.h): a stream of independent pricing calls with several launches in flight on one GPU.

The reference's 20-run validation launches, synchronises and copies once per run (src/2_option_pricing.cu:225-302); the
pipe enqueues the runs alternating over hardware queues and synchronises once.  Every request is one full pricing call,
so the sums must equal those of separate hw_zbc_moments calls BIT FOR BIT -- whatever the number of queues, the streams
the tuning step picked, the batch length or the order of the calls."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEEDS = [1234 + 12345 * k for k in range(20)]              # 2_option:229


@pytest.fixture(scope="module")
def market(hwmc_mod):
    p = hwmc_mod.default_params("readme")
    P, f = hwmc_mod.analytic_curve(p)
    e = hwmc_mod.HullWhiteEngine(p, device=0)
    e.load_market_data(P, f)
    yield p, P, f, e
    e.close()


@pytest.mark.parametrize("queues", [1, 2, 3, 4])
def test_batch_equals_separate_calls(hwmc_mod, market, queues):
    p, P, f, e = market
    with hwmc_mod.HullWhitePipeline(p, device=0, queues=queues) as pipe:
        pipe.load_market_data(P, f)
        for n in (1000, 70_001, 1 << 20):                                   # the last one: two pairs per thread
            got, ms = pipe.simulate_ZBC_batch(n, SEEDS, return_ms=True)
            want = np.array([e.simulate_ZBC_control_variate(n, s) for s in SEEDS])
            assert got.shape == (20, 5) and ms > 0.0
            assert np.array_equal(got, want), (queues, n)
        # the single-launch form of the same 20 runs (hw_zbc_moments_runs) agrees too
        assert np.array_equal(pipe.simulate_ZBC_batch(1 << 18, SEEDS), e.simulate_ZBC_runs(1 << 18, 1234, n_runs=20, seed_stride=12345))
        # batches shorter than the number of queues, an empty one, offsets, the reference stepper
        assert pipe.simulate_ZBC_batch(5000, []).shape == (0, 5)
        for k in (1, 2, 3, 5):
            got = pipe.simulate_ZBC_batch(5000, SEEDS[:k], path_offset=(1 << 32) - 77)
            want = np.array([e.simulate_ZBC_control_variate(5000, s, path_offset=(1 << 32) - 77) for s in SEEDS[:k]])
            assert np.array_equal(got, want), k
        ref = hwmc_mod.HW_STEPPER_REFERENCE
        assert np.array_equal(pipe.simulate_ZBC_batch(3000, SEEDS[:4], stepper=ref),
                              np.array([e.simulate_ZBC_control_variate(3000, s, stepper=ref) for s in SEEDS[:4]]))


def test_errors_leave_the_pipe_usable(hwmc_mod, market):
    p, P, f, e = market
    with pytest.raises(hwmc_mod.HwError):
        hwmc_mod.HullWhitePipeline(p, device=0, queues=0)
    with pytest.raises(hwmc_mod.HwError):
        hwmc_mod.HullWhitePipeline(p, device=0, queues=5)
    with hwmc_mod.HullWhitePipeline(p, device=0, queues=2) as pipe:
        with pytest.raises(hwmc_mod.HwError) as ei:                          # no market data yet
            pipe.simulate_ZBC_batch(1000, SEEDS[:2])
        assert "market" in str(ei.value)
        pipe.load_market_data(P, f)
        first = pipe.simulate_ZBC_batch(4096, SEEDS[:6])
        with pytest.raises(hwmc_mod.HwError) as ei:                          # S2 > T_final: refused before anything is enqueued
            pipe.simulate_ZBC_batch(4096, SEEDS[:6], S1=5.0, S2=11.0)
        assert "S1 < S2" in str(ei.value)
        with pytest.raises(hwmc_mod.HwError):                                # found by the launch path (f32-only stepper is fine; bad stepper id)
            pipe.simulate_ZBC_batch(4096, SEEDS[:6], stepper=7)
        assert np.array_equal(pipe.simulate_ZBC_batch(4096, SEEDS[:6]), first)
        assert pipe.tune(1 << 16) > 0.0                                      # explicit tuning on another workload
        assert np.array_equal(pipe.simulate_ZBC_batch(4096, SEEDS[:6]), first)


def test_two_queues_alternate_and_overlap(hwmc_mod, market):
    """The point of the pipe, asserted WITHOUT a wall clock (a noisy neighbour must not turn the correctness
    suite red): 40 pricing calls of 2^20 pairs through two queues
      * return the sums of the same calls made synchronously one after another, bit for bit;
      * alternate over the two queues -- request i on queue i % 2, twenty launches on each, the queues on two DISTINCT
        candidate streams (hw_pipe_queue) --, every queue's launches in request order and never overlapping each other;
      * are in flight together: on the common event time axis of hw_profile_spans a launch of one queue starts before the
        previous request's launch on the other queue has ended (streams that share a hardware queue serialise completely --
        zero such pairs -- and picking streams that do not is what hw_pipe_tune is for).
    How much wall time that saves is a throughput claim: tests/test_perf_claims.py, -m perf."""
    p, P, f, e = market
    n, seeds = 1 << 20, [7 + k for k in range(40)]
    with hwmc_mod.HullWhitePipeline(p, device=0, queues=2) as pipe:
        pipe.load_market_data(P, f)
        assert pipe.tune(n) > 0.0
        (q0, s0, nq), (q1, s1, _) = pipe.queue(0), pipe.queue(1)
        assert nq == 2 and s0 != s1 and 0 <= s0 < 8 and 0 <= s1 < 8
        with pytest.raises(hwmc_mod.HwError):
            pipe.queue(2)
        pipe.simulate_ZBC_batch(n, seeds)                                    # one untimed batch: first-launch effects out of the picture
        q0.profile_enable(len(seeds))
        q1.profile_enable(len(seeds))
        got = pipe.simulate_ZBC_batch(n, seeds)
        a0, b0 = q0.profile_spans(ref=q0)
        a1, b1 = q1.profile_spans(ref=q0)
        assert len(q0.profile_read()) == 20 and len(q1.profile_read()) == 20 and len(a0) == len(a1) == 20
        want = np.array([e.simulate_ZBC_control_variate(n, s) for s in seeds])
        assert np.array_equal(got, want)
        for a, b in ((a0, b0), (a1, b1)):                                    # one queue: in order, one launch at a time
            assert np.all(b > a) and np.all(a[1:] >= b[:-1])
        # request 2k ran on queue 0, request 2k+1 on queue 1: consecutive requests on different queues overlap
        overlaps = int(np.sum(a1 < b0)) + int(np.sum(a0[1:] < b1[:-1]))
        assert overlaps >= 1, "no launch of one queue ever started before the other queue's previous launch ended: the two streams serialise"
        print("two queues: %d of 39 consecutive launches overlapped; streams %d and %d" % (overlaps, s0, s1))


def test_automatic_tuning_on_the_first_request_returns_the_right_sums(hwmc_mod, market):
    """An untuned pipe tunes inside its first pricing call -- on a COPY of the first request capped at 2^20 pairs, 8 launches
    per trial (tuning on a 2^24-pair request cost the first call ~a thousand launches of it) -- and the call returns
    the sums of the REAL request.  (What the tuning costs in wall time is asserted by the -m perf test below: a clock ramp or a
    noisy neighbour must not fail the correctness suite)"""
    p, P, f, e = market
    n = 1 << 24
    with hwmc_mod.HullWhitePipeline(p, device=0, queues=2) as pipe:
        pipe.load_market_data(P, f)
        got = pipe.simulate_ZBC_batch(n, SEEDS[:2])
        assert np.array_equal(got, np.array([e.simulate_ZBC_control_variate(n, s) for s in SEEDS[:2]]))
        assert np.array_equal(got, pipe.simulate_ZBC_batch(n, SEEDS[:2]))        # tuned now: same sums


@pytest.mark.perf
def test_automatic_tuning_is_bounded_whatever_the_first_request(hwmc_mod, market):
    """The first call of an untuned pipe on a 2^24-pair request must cost about what tuning on 2^20 pairs costs, not seconds."""
    p, P, f, e = market
    n = 1 << 24
    with hwmc_mod.HullWhitePipeline(p, device=0, queues=2) as pipe:
        pipe.load_market_data(P, f)
        t0 = time.perf_counter()
        pipe.simulate_ZBC_batch(n, SEEDS[:2])
        first_call_s = time.perf_counter() - t0
    # measured against an explicit tuning pass on a 2^20-pair request in the same process: 100 ms clock ramp + 28 pairs x 2 reps
    # x 8 launches of <= 2^20 pairs + the two real calls ~ 0.25 s; uncapped it was 28 x 2 x 16 launches x 2.7 ms = 2.4 s + the
    # ramp, i.e. ~8 x the explicit pass
    with hwmc_mod.HullWhitePipeline(p, device=0, queues=2) as pipe:
        pipe.load_market_data(P, f)
        t0 = time.perf_counter()
        pipe.tune(1 << 20)
        explicit_s = time.perf_counter() - t0
    assert first_call_s < 2.5 * explicit_s + 0.05, (first_call_s, explicit_s)
