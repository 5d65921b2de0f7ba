#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Hull-White Monte Carlo hot path on MI355X.

Metric (BASELINE.json): M paths/sec of the ZBC option kernel (2^20 antithetic pairs per GPU,
1000-step grid over T=10 -> 500 executed steps to S1=5, optimal-beta control variate),
throughput defined exactly as the reference does: (2*N_PATHS / ms) / 1000
(src/2_option_pricing.cu:198).  One "step" = one pricing call = one launch of the path kernel
over the whole batch + the reduction of its five moment sums; the K timed steps use K different
seeds, i.e. they ARE the reference's 20-run statistical validation (2_option:210-468), from
which the 95 % CI width of the metric is computed.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Pipelining: the K steps are independent pricing batches, so each GPU alternates them over S contexts (S HIP
streams, --streams, default 2): the ~15 us a 2^20-pair launch loses to launch latency and fill/drain
(tools/size_scan.py) is filled by the next batch's waves.  Every batch is still one full launch; the sums are
bit-identical to the single-stream run (tools/overlap_probe.py) and the JSON line also carries the
single-stream figure ("serial").

N > 1: one process per GPU; rank g prices global pairs [g*n, (g+1)*n) of an N*n-pair run (weak
scaling), and the five moment sums of every step are all-reduced over RCCL inside the timed region:
by default as ONE message of 5*K doubles after the last launch (--allreduce-every B sends one message
per B steps from a stream of its own; measured with a one-rank communicator that costs 3-6 % because
RCCL's stream lands on one of the two hardware queues the launches alternate over).  Rank 0 prints ONE
JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# VALU issue model of the path kernel's inner loop, per Philox block = 4 time steps of one pair
# (instruction counts from the gfx950 disassembly -- tools/loop_census.py --, issue cycles per wave64 instruction per SIMD
#  measured by tools/valu_rates.hip on MI355X: full rate 2, half rate 4, quarter rate 8).
VALU_MODEL = {
    "superposed": {"full": 30, "half": 21, "quarter": 8},   # 14 bitop3,4 xor,4 fmamk,2 mul,4 fma/fmac,2 add | 16 mad_u64,4 cvt,1 pk_fma | 2 log,2 sqrt,2 sin,2 cos
    "reference": {"full": 64, "half": 20, "quarter": 8},    # + per step: sG mul, 2 add, 2 fma, 2 add, 2 mul/fma pairs on two legs
    # double stepper: per-step fmac_f64 + add_f64 on converted normals:
    # 14 bitop3,4 xor,4 fmamk,4 mul | 16 mad_u64,4 cvt_u32,4 cvt_f64,4 fmac_f64,4 add_f64 | 8 transcendentals
    "superposed_f64": {"full": 26, "half": 32, "quarter": 8},
}
# Issue costs measured on MI355X at 8 waves/SIMD (tools/valu_rates.hip, tools/valu_patterns.hip; profiles/):
# "full-rate" ops sustain one per 2.3 cycles (two waves issue side by side; one wave alone: 4.7),
# v_mad_u64_u32 4.45, v_cvt_f32_u32 4.24, transcendentals 8.1 -- each class measured alone; in the mixed loop part of
# v_cvt hides under single-rate ops (tools/coissue_probe.hip) and the kernel runs at the nominal 2/4/8 model.
MEASURED_ISSUE = {"full": 2.3, "half": 4.4, "quarter": 8.1}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
V100_PUBLISHED_MPATHS = 1064.0  # BASELINE.md section 1 (reference README.md:44,162; one V100)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths-log2", type=int, default=20, help="antithetic pairs per GPU (log2)")
    ap.add_argument("--grid-steps", type=int, default=1000)
    ap.add_argument("--stepper", default="superposed", choices=["superposed", "reference"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--profile", default="head", choices=["head", "readme"])
    ap.add_argument("--block", type=int, default=0, help="threads per workgroup (0 = engine default)")
    ap.add_argument("--streams", type=int, default=2,
                    help="contexts (HIP streams) per GPU the independent steps alternate over; 1 = strictly serial")
    ap.add_argument("--allreduce-every", type=int, default=0,
                    help="N > 1: all-reduce the sums of this many consecutive steps as one message; 0 (default) = ONE "
                         "all-reduce of all K steps' sums after the last launch, inside the timed region -- the K steps "
                         "are the reference's 20-run validation, whose statistics need the sums only at the end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--spinup-ms", type=float, default=400.0,
                    help="untimed launches before the W warm-up steps so the timed region runs at the "
                         "clock the chip holds under sustained load (DVFS ramps over the first ~100 ms)")
    return ap.parse_args()


def cpu_baseline(P, f, profile, grid_steps, seconds):
    """The oracle's OpenMP loop (same stepper, same Philox stream) on this host's cores, on a
    bounded sample of the same workload; throughput is per path so it compares directly."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hw_oracle
    o = hw_oracle.Oracle(profile, n_steps=grid_steps)
    cores, how = hw_oracle.usable_cores(o.omp_max_threads())
    t0 = time.perf_counter()
    o.zbc_moments(P, f, 1 << 13, 1, n_threads=cores)     # calibration
    rate = (1 << 13) / max(time.perf_counter() - t0, 1e-6)
    n = int(min(max(rate * seconds, 1 << 13), 1 << 25))
    t0 = time.perf_counter()
    o.zbc_moments(P, f, n, 2, n_threads=cores)
    dt = time.perf_counter() - t0
    n1 = max(n // (4 * cores), 1 << 10)
    t0 = time.perf_counter()
    o.zbc_moments(P, f, n1, 3, n_threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": 2 * n / dt / 1e6, "unit": "M paths/s", "cores": cores, "cores_how": how, "kind": "port",
            "sample": "%d antithetic pairs x %d executed steps, OpenMP %d threads, %.1f s (oracle/hw_oracle.c)"
                      % (n, o.steps_to(5.0), cores, dt),
            "single_thread_value": 2 * n1 / dt1 / 1e6,
            "single_thread_sample": "%d pairs, %.1f s" % (n1, dt1)}


def main():
    a = parse()
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)  # a default run takes well under a minute: never sit in a hang
    import torch                                       # first: libhw_hip.so then shares torch's HIP runtime
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # Rehearsal mode for a box with fewer GPUs than ranks (HWMC_BENCH_REHEARSAL=1): every rank uses
    # device 0 and the all-reduce goes through gloo on the host.  It exercises the N>1 control flow
    # only; its numbers are NOT a benchmark.  The real path is nccl (= RCCL) with one GPU per rank.
    rehearsal = os.environ.get("HWMC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HWMC_BENCH_FORCE_RCCL=1: run the RCCL code path (communicator, per-step all-reduce ordered after the step's
    # stream, barriers) even in a single process -- the only way to exercise it on a one-GPU box.
    dist_on = world > 1 or os.environ.get("HWMC_BENCH_FORCE_RCCL") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import numpy as np
    import hwmc
    n_pairs = 1 << a.paths_log2
    p = hwmc.default_params(a.profile, n_steps=a.grid_steps, precision=a.precision)
    P, f = hwmc.analytic_curve(p)                      # synthetic market input (exact curve)
    S = max(1, min(a.streams, 8))
    engines = []
    for _ in range(S):                                 # one context = one stream + its own partial-sum workspace
        eng = hwmc.HullWhiteEngine(p, device=local_rank)
        eng.load_market_data(P, f)
        if a.block:
            eng.set_launch(block_size=a.block)
        engines.append(eng)
    e = engines[0]
    stepper = hwmc.HW_STEPPER_SUPERPOSED if a.stepper == "superposed" else hwmc.HW_STEPPER_REFERENCE
    K, W = a.steps, a.warmup
    out = torch.zeros((K + W, 5), dtype=torch.float64, device=dev)
    # Explicit side streams: step i runs on streams[i % S], made torch's CURRENT stream while it is issued, so
    # the kernels go to it through the C ABI, torch's events time it and RCCL orders that step's all-reduce
    # after it.  (A NULL handle would mean "the context's own stream" to the C ABI.)
    # HIP multiplexes its streams onto a few hardware queues (4 here) and two streams that share a queue
    # serialise (tools/queue_map_probe.py: which streams share one depends on the creation history of the whole
    # process).  So take 8 candidate streams and keep, greedily, those that run the workload fastest together.
    cands = [torch.cuda.Stream(device=dev) for _ in range(8 if S > 1 else 1)]
    assert all(st.cuda_stream != 0 for st in cands)
    pairing = None
    if S > 1:
        cal_out = torch.zeros((8, 5), dtype=torch.float64, device=dev)

        def trial(chosen):
            best = float("inf")
            for _ in range(3):
                torch.cuda.synchronize()
                tc = time.perf_counter()
                for j in range(8):
                    engines[j % len(chosen)].enqueue_ZBC(cal_out[j].data_ptr(), n_pairs, 7 + j, stepper=stepper,
                                                         stream=cands[chosen[j % len(chosen)]].cuda_stream)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - tc) * 1e3 / 8)
            return best

        for _ in range(40):                                 # clocks up before comparing anything
            trial([0])
        chosen, table = [0], {}
        while len(chosen) < S:
            scores = {j: trial(chosen + [j]) for j in range(len(cands)) if j not in chosen}
            table[len(chosen) + 1] = {str(j): round(v, 4) for j, v in scores.items()}
            chosen.append(min(scores, key=scores.get))
        streams = [cands[j] for j in chosen]
        pairing = {"candidates": len(cands), "chosen": chosen, "ms_per_step_when_added": table,
                   "ms_per_step_single_stream": round(trial([0]), 4)}
    else:
        streams = cands
    torch.cuda.synchronize()
    torch.cuda.set_stream(streams[0])
    # weak scaling: an (N * n_pairs)-pair run, rank g owns the contiguous shard hwmc.parallel.shard gives it
    offset, shard_n = hwmc.parallel.shard(n_pairs * world, rank, world)
    assert shard_n == n_pairs

    class _Done:
        def wait(self):
            return None

    def enqueue(i, seed, dst, n_streams=S, which=stepper):
        k = i % n_streams
        engines[k].enqueue_ZBC(dst.data_ptr(), n_pairs, seed, path_offset=offset, stepper=which,
                               stream=streams[k].cuda_stream)
        return k

    # Collectives: the sums of B consecutive steps (--allreduce-every, rows of `out` are contiguous) go out as ONE
    # all-reduce of 5*B doubles.  It is issued from a stream of its own that waits on an event of every compute
    # stream, so no compute stream ever waits for another one or for RCCL.
    B = max(1, a.allreduce_every) if a.allreduce_every > 0 else (K + W)
    comm_stream = torch.cuda.Stream(device=dev)
    pending = {"first": None}

    def flush(i_last):
        i0, pending["first"] = pending["first"], None
        if i0 is None or not dist_on:
            return None
        if rehearsal:                                                   # host round trip, synchronous
            torch.cuda.synchronize()
            tmp = out[i0:i_last + 1].cpu()
            dist.all_reduce(tmp)
            out[i0:i_last + 1].copy_(tmp)
            return _Done()
        for st in streams:
            comm_stream.wait_stream(st)                                 # event record + wait: the compute streams run on
        with torch.cuda.stream(comm_stream):
            return hwmc.parallel.allreduce_sums(out[i0:i_last + 1], async_op=True, force=True)

    def step(i, seed, last=False):
        enqueue(i, seed, out[i])
        if pending["first"] is None:
            pending["first"] = i
        if last or i - pending["first"] + 1 >= B:
            return flush(i)
        return None

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def join_streams():                                                 # streams[0] waits for the others
        for st in streams[1:]:
            streams[0].wait_stream(st)

    if dist_on:                                                        # build the communicator outside any timed region
        warm = torch.zeros(5, dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(warm)
        fence()

    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < a.spinup_ms:      # untimed: bring the clocks up
        for j in range(8):
            enqueue(j, 7, out[j % (K + W)])
        torch.cuda.synchronize()
    for i in range(W):
        w = step(i, 900 + i, last=(i == W - 1))
        if w is not None:
            w.wait()
    fence()
    # One HIP event pair brackets the K timed launches: recorded on streams[0] before the first launch and,
    # after streams[0] has been made to wait for the other streams, behind the last one.  Per-launch event
    # pairs are taken in a second, untimed pass below.
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    works = []
    t0 = time.perf_counter()
    ev0.record(streams[0])
    for i in range(K):
        works.append(step(W + i, 1234 + 12345 * i, last=(i == K - 1)))   # seeds as 2_option:229: base + run*12345
    join_streams()
    ev1.record(streams[0])
    for w in works:
        if w is not None:
            w.wait()
    torch.cuda.synchronize()                               # every local launch AND every all-reduce has completed, i.e.
    elapsed = time.perf_counter() - t0                     # every rank has contributed all its K steps; MAX over ranks below
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize()
    region_ms = ev0.elapsed_time(ev1)
    sums = out[W:].cpu().numpy()                           # the K all-reduced moment vectors of the timed steps
    scratch = torch.zeros((K, 5), dtype=torch.float64, device=dev)
    # untimed pass 1: the same K launches in the same alternation with a hipEvent pair tight around every path
    # kernel -- the per-launch durations rocprofv3's kernel trace reports for the timed region (with S > 1
    # two launches share the chip for part of their life, so each one lasts longer than it would alone)
    for eng in engines:
        eng.profile_enable(K)
    for i in range(K):
        enqueue(i, 1234 + 12345 * i, scratch[i])
    torch.cuda.synchronize()
    kernel_ms = np.concatenate([eng.profile_read(K) for eng in engines])
    assert kernel_ms.size == K
    # untimed pass 2: the same K launches strictly one after another on ONE stream: isolated kernel durations
    # and the wall-clock rate without pipelining
    e.profile_enable(K)
    for i in range(K):
        enqueue(i, 1234 + 12345 * i, scratch[i], n_streams=1)
    torch.cuda.synchronize()
    kernel_ms_isolated = e.profile_read(K)
    for eng in engines:
        eng.profile_enable(0)                              # event pairs off: they cost queue bubbles
    torch.cuda.synchronize()
    ts = time.perf_counter()
    for i in range(K):
        enqueue(i, 1234 + 12345 * i, scratch[i], n_streams=1)
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - ts) * 1e3 / K
    assert world > 1 or np.array_equal(scratch.cpu().numpy(), sums), "the stream count changed the sums"
    clock_mhz = e.profile_clock_mhz() if a.stepper == "superposed" else None
    # For the record, the OTHER stepper on the same workload (untimed w.r.t. the headline; N = 1 only): the
    # reference's per-step arithmetic verbatim vs the superposed recursion, so both numbers sit in one line.
    alt = None
    if world == 1:
        other = hwmc.HW_STEPPER_REFERENCE if a.stepper == "superposed" else hwmc.HW_STEPPER_SUPERPOSED
        if not (other == hwmc.HW_STEPPER_REFERENCE and a.precision != "f32"):
            for i in range(3):
                enqueue(i, 5, scratch[i % K], which=other)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for i in range(K):
                enqueue(i, 1234 + 12345 * i, scratch[i], which=other)
            torch.cuda.synchronize()
            alt_ms = (time.perf_counter() - ta) * 1e3 / K
            alt = {"stepper": "reference" if other == hwmc.HW_STEPPER_REFERENCE else "superposed",
                   "ms_per_step": alt_ms, "value": (2 * n_pairs / alt_ms) / 1000.0, "unit": "M paths/s"}
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_total = 2 * n_pairs * world
        prices = [hwmc.cv_price(s, n_total, float(P[-1])) for s in sums]
        st = hwmc.run_statistics([r["price"] for r in prices])
        ms_per_step = elapsed * 1e3 / K
        value = (n_total / ms_per_step) / 1000.0
        steps_exec = int(round(5.0 * a.grid_steps / 10.0))
        info = e.device_info()
        # Path-kernel duration.  kiso: hipEvent pairs tight around each of K launches issued strictly one after another
        # (untimed pass 2) = the kernel alone on the chip; rocprofv3's kernel-trace average for this command run with
        # --streams 1 reports the same quantity (profiles/rNN_bench_streams1_kernel_stats.csv) and roofline.achieved
        # uses it.  kavg: the same K launches replayed in the timed region's alternation over S streams (pass 1): two
        # dispatches then share the chip, so each one lasts up to twice as long as it occupies the chip for; the
        # comparable rocprofv3 figure for the default command is the union of the dispatch intervals per launch
        # (profiles/rNN_pmc_summary.json, kernel_trace.chip_time_per_launch_us) against timed_region_ms_per_step.
        kavg = float(np.mean(kernel_ms))
        kiso = float(np.mean(kernel_ms_isolated))
        region_per_step = region_ms / K
        geo_blocks = min((n_pairs + (a.block or 512) - 1) // (a.block or 512), 1 << 20)   # one 256-thread workgroup per 256 pairs (csrc/hw_engine.hip geometry())
        algo_bytes = geo_blocks * 5 * 8 + 5 * 8            # partial rows written + the 5 sums (stateless Philox: no state traffic)
        achieved = algo_bytes / (kiso * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc) and a.stepper == "superposed" and a.paths_log2 == 20 and a.grid_steps == 1000:
            with open(pmc) as fh:
                traffic = json.load(fh)["hbm_traffic_bytes_per_launch"]["corrected_bytes"]
            traffic_src = "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH_SIZE x2 on gfx950)"
        vm = VALU_MODEL["superposed_f64" if (a.stepper == "superposed" and a.precision == "f64") else a.stepper]
        cyc_per_block = 2 * vm["full"] + 4 * vm["half"] + 8 * vm["quarter"]
        waves = (n_pairs + 63) // 64
        simd_cycles = waves * (steps_exec / 4.0) * cyc_per_block
        simd_hz = info["compute_units"] * 4 * info["clock_mhz"] * 1e6
        valu_frac = simd_cycles / (simd_hz * region_per_step * 1e-3)        # the chip over the timed region
        valu_frac_iso = simd_cycles / (simd_hz * kiso * 1e-3)               # one launch alone on the chip
        achieved_cyc = region_per_step * 1e-3 * (clock_mhz or info["clock_mhz"]) * 1e6 * info["compute_units"] * 4 / (waves * steps_exec / 4.0)
        line = {
            "metric": "M paths/sec (ZBC option, 1000 steps)", "value": round(value, 2), "unit": "M paths/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / V100_PUBLISHED_MPATHS, 3),
            "baseline": {"value": V100_PUBLISHED_MPATHS, "unit": "M paths/s", "hardware": "1x Tesla V100",
                         "source": "reference README.md:44,162 (single un-warmed launch)"},
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "Q2b option_pricing: 2^%d antithetic pairs per GPU x %d-step grid (%d executed to S1=5), "
                                   "optimal-beta control variate, %d x MI355X" % (a.paths_log2, a.grid_steps, steps_exec, world),
                       "pairs_per_gpu": n_pairs, "effective_paths": n_total, "grid_steps": a.grid_steps,
                       "steps_executed": steps_exec, "stepper": a.stepper, "theta_profile": a.profile,
                       "rng": "Philox4x32-10 (rocRAND layout), stateless, subsequence = global pair id",
                       "parallelism": ("paths sharded by global id; the 5 f64 moment sums of every step all-reduced over RCCL, %d steps per message" % B)
                                      if world > 1 else "single GPU",
                       "streams_per_gpu": S, "stream_selection": pairing,
                       "pipelining": ("the K independent steps alternate over %d contexts/streams per GPU; every step is one full launch" % S)
                                     if S > 1 else "none: steps run strictly one after another"},
            "thread_steps_per_sec": n_pairs * world * steps_exec / (ms_per_step * 1e-3),
            "price_mean": st["mean"], "ci95_width": st["ci95_width"], "ci95_halfwidth_single_run": prices[0]["ci95_halfwidth"],
            "beta_mean": float(np.mean([r["beta"] for r in prices])),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "zbc_%s_kernel" % a.stepper, "kernel_ms_avg": kiso, "kernel_ms_min": float(np.min(kernel_ms_isolated)),
                         "kernel_ms_avg_basis": "K launches back to back on one stream, hipEvent pair around each (matches rocprofv3 --stats of this command with --streams 1)",
                         "kernel_ms_avg_overlapped_replay": kavg, "kernel_ms_isolated_avg": kiso,
                         "timed_region_ms_per_step_hip_events": region_per_step,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "stateless Philox: the path kernel has ~0 algorithmic HBM traffic; the binding roof is VALU issue (see valu)"},
            "valu": {"bound": "valu-issue", "model": "serial issue: 2 / 4 / 8 cycles per single-rate / multiply-convert-f64 / transcendental wave64 instruction per SIMD",
                     "issue_cycles_per_wave_per_4_steps": cyc_per_block, "instr_per_4_steps": vm,
                     "achieved_cycles_per_wave_per_4_steps": achieved_cyc,
                     "clock_mhz_nominal": info["clock_mhz"], "clock_mhz_in_kernel": clock_mhz,
                     "frac_of_nominal_clock_peak": valu_frac,
                     "frac_at_measured_clock": (valu_frac * info["clock_mhz"] / clock_mhz) if clock_mhz else None,
                     "frac_of_nominal_clock_peak_isolated_launch": valu_frac_iso,
                     "note": "fractions compare the model's cycles with the chip time per step of the timed region (isolated: with one "
                             "launch alone on the chip); they can pass 1 because part of v_cvt_f32_u32 hides under single-rate ops "
                             "(profiles: coissue probe).  Single-class issue rates measured alone: %s cycles." % MEASURED_ISSUE},
            "kernel_Mpaths_per_sec_isolated_launch": (2 * n_pairs / kiso) / 1000.0,
            "serial": {"streams_per_gpu": 1, "ms_per_step": serial_ms, "value": (2 * n_pairs / serial_ms) / 1000.0,
                       "unit": "M paths/s per GPU", "note": "same K launches strictly one after another on one stream, no all-reduce"},
            "other_stepper": alt,
            "device": info["name"],
        }
        if rehearsal:
            line["rehearsal"] = "gloo, all ranks on device 0 -- control-flow check only, not a measurement"
        assert 0.0 < st["mean"] < 0.1 and all(np.isfinite(s).all() for s in sums), "implausible reduced sums"
        if os.environ.get("HWMC_BENCH_TRACE"):
            print("bench: GPU legs done", file=sys.stderr, flush=True)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(P, f, a.profile, a.grid_steps, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    for eng in engines:
        eng.close()
    if os.environ.get("HWMC_BENCH_TRACE"):
        print("bench: contexts closed", file=sys.stderr, flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
    if os.environ.get("HWMC_BENCH_TRACE"):
        print("bench: main returned", file=sys.stderr, flush=True)
