#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Hull-White Monte Carlo hot path on MI355X.

Metric (BASELINE.json): M paths/sec of the ZBC option kernel (1000-step grid over T=10 -> 500 executed steps to S1=5,
optimal-beta control variate), throughput defined exactly as the reference does: (2*N_PATHS / ms) / 1000
(src/2_option_pricing.cu:198).  One "step" = one pricing call = one launch of the path kernel over the whole batch +
the reduction of its five moment sums (+ for N > 1 ONE all-reduce of those five doubles over RCCL); the K timed steps
use K different seeds, i.e. they ARE the reference's 20-run statistical validation (2_option:210-468), from which
the 95 % CI width of the metric is computed.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload.  N = 1: BASELINE configs[1], 2^20 antithetic pairs per step.  N > 1: BASELINE configs[3], STRONG scaling of a
2^24-pair pricing call (--total-log2): rank g prices the global pairs [g*2^24/N, (g+1)*2^24/N) of every step, and the five
sums of every step are all-reduced over RCCL -- one collective per pricing call (--allreduce-every 1, the default),
issued right behind the step's kernels: torch's RCCL backend runs it on a stream of its own that waits for an event of the
step's compute stream, so no compute stream ever waits for RCCL.  (--scaling weak keeps 2^20 pairs per GPU; --allreduce-every 0 sends ONE message for all K steps.)
Because the sample set depends only on (seed, global pair id), the reduced sums do not depend on N.

Timing.  EXACTLY K steps are timed between barrier + torch.cuda.synchronize() on both sides, MAX over ranks; that
region is repeated (fresh barriers each time) and `value` is the MEDIAN region, with the spread beside it: at 2^20 pairs a
region lasts only K * 0.18 ms, so by default (--regions 0) it is repeated until the timed regions total >= 0.5 s
(`timed_region_s` in the line; ~145 regions at N = 1), which also makes the timed work visible to an outside sampler.

Pipelining.  The K steps are independent pricing batches, so each GPU alternates them over S contexts (S HIP streams,
--streams, default 2): the drain of a launch's last waves (~11 us at 2^20 pairs) and the ~4 us it spends outside its waves
(dispatch, end-of-kernel release) are covered by the next batch.  Every batch is still one full launch; the sums are bit-identical to the
single-stream run, and the line carries the strictly serial rate ("serial") and the kernel alone on the chip
("kernel_Mpaths_per_sec_isolated_launch", which is what rocprofv3 --stats reports for --streams 1) next to `value`.

Side measurements in the same line (untimed, N = 1; --no-extras drops them): `long_region` (10 K steps between two
synchronisations: what starting and draining the pipeline once per K steps costs), `library_pipeline` (the same K calls
through the C ABI's own hw_pipe_zbc_moments: one function call, no torch), `other_stepper`, `runs20_single_launch`,
`q1_curve_gpu`, `q3_vega_gpu` (the reference's other two published rates, README.md:161-163), `q3_fd_vega_gpu` (the
three-scenario finite-difference launch, 3_sens:400-446), `size_table` (2^20..2^26 pairs, kernel alone), `sync_call` (host-to-host
time of a synchronous hw_zbc_moments call), `stress` (BASELINE configs[4] on one GPU: 2^26 pairs x 4000-step grid, float and double
engine, ZBC stepper and curve kernel, each with its HBM and VALU fraction), `strong_scaling_reference`;
`cpu_baseline` = the oracle on the host cores (the only use of oracle/ here).

Self-verifying line.  Every timed side launch also checks what it computed (`checks`: the price / Vega / curve it returned
against the Hull-White closed forms within 4 standard errors, float against double to 1e-5, bit-identity where two routes must
give the same sums, the committed one-GPU constants of <package>/reference_sums.json to 1e-13) and the line ends with `claims`:
every rate README.md's table quotes, with its ms and its HBM / VALU fractions, in <= 1.5 KB -- the LAST key, so that a record
which keeps only the tail of this (16 KB) line still carries all of them.  A failed check: the line is printed, exit code 3.
For N > 1 the last key is `multi_gpu`, ending in `self_check` (hwmc.parallel.self_check): ranks seen == --gpus, sums_step0
equal to the committed one-GPU sums of the configuration (else exit code 3), strong scaling against the committed one-GPU
time and against the prediction of DESIGN.md section 6.

VALU bound.  `valu.bound_cycles_per_block` is MEASURED, not modelled: tools/mix_bound.hip times each production loop's own
instruction stream with its data dependencies removed (tools/gen_mix_bound.py) in shader-clock cycles: the rate at which a
SIMD issues that instruction mix in the production ORDER when nothing waits for a result.  It is the fastest of the orderings
and occupancies tried, not a proven floor (the provable one, `exclusive_classes_floor`, is lower); valu.frac = bound / achieved
is the share of a launch's SIMD time that the loop's issue rate accounts for -- the rest is prologue, epilogue, drain and
dependency stalls.  Loops with a twin: the float ZBC/Vega loop, the double ZBC loop, the curve loop (P = 1 pair per thread on
the 1000-step grid; P = 2 on the 4000-step grid, float and double).  bench.py runs the probe live (a child process, before this process touches the GPU) when
tools/bin/mix_bound exists, else reads the newest profiles/rNN_mix_bound.json.
"""
import argparse
import glob
import json
import math
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Multi-process GPU work on this pool needs dmabuf IPC (RCCL's buffer exchange fails with "hipIpcGetMemHandle: invalid argument"
# under the legacy mode); the variable is read when the HSA runtime starts, i.e. before torch / libhw_hip.so touch the GPU.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# VALU issue model of the path kernel's inner loop, per Philox block = 4 time steps of one pair
# (instruction counts from the gfx950 disassembly -- tools/loop_census.py, pinned by tests/test_kernel_census.py --,
#  issue cycles per wave64 instruction per SIMD measured by tools/valu_rates.hip on MI355X: full rate 2, half rate 4,
#  quarter rate 8).
VALU_MODEL = {
    "superposed": {"full": 32, "half": 20, "quarter": 8},   # 14 bitop3,4 xor,4 fmamk,2 mul,6 fma/fmac,2 add | 16 mad_u64,4 cvt | 2 log,2 sqrt,2 sin,2 cos
    # reference stepper: per step and leg v_fma (d +- sig_st g), v_fmac (r' = r e + t), v_add (r + r'), v_fmac (I += half_dt (r + r'));
    # per block + 4 products for the normals: 14 bitop3,4 xor,4 fmamk,4 mul,8 fma,16 fmac,8 add | 16 mad_u64,4 cvt | 8 transcendentals
    "reference": {"full": 58, "half": 20, "quarter": 8},
    # double stepper, (y, W) block form: 14 bitop3,4 xor,4 fmamk,3 mul,2 add,2 fma,1 fmac | 16 mad_u64,4 cvt_u32,3 cvt_f64,2 fmac_f64,1 add_f64 | 8 transcendentals
    "superposed_f64": {"full": 30, "half": 26, "quarter": 8},
}
MEASURED_ISSUE = {"full": 2.3, "half": 4.4, "quarter": 8.1}
V100_PUBLISHED_Q1_MPATHS = 391.0   # reference README.md:161 (simulate_zcb, 2 x 2^20 paths x 1000 steps)
V100_PUBLISHED_Q3_MPATHS = 509.0   # reference README.md:163 (simulate_sensitivity, 2^20 paths; src/3_sensitivity_analysis.cu:266)
TIMED_TOTAL_S = 0.5                # --regions 0: repeat the K-step region until the timed regions total this much
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
V100_PUBLISHED_MPATHS = 1064.0  # BASELINE.md section 1 (reference README.md:44,162; one V100)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=0,
                    help="how many times the timed K-step region is repeated (median reported); 0 = as many as make the timed "
                         "regions total >= 0.5 s (at least 7, at most 400)")
    ap.add_argument("--paths-log2", type=int, default=20, help="antithetic pairs per step: per GPU for N = 1 and for --scaling weak (log2)")
    ap.add_argument("--total-log2", type=int, default=24, help="N > 1, --scaling strong: pairs per step over ALL GPUs (log2); BASELINE configs[3] = 24")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--grid-steps", type=int, default=1000)
    ap.add_argument("--stepper", default="superposed", choices=["superposed", "reference"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--profile", default="head", choices=["head", "readme"])
    ap.add_argument("--block", type=int, default=0, help="threads per workgroup (0 = engine default)")
    ap.add_argument("--streams", type=int, default=2,
                    help="contexts (HIP streams) per GPU the independent steps alternate over; 1 = strictly serial")
    ap.add_argument("--allreduce-every", type=int, default=1,
                    help="N > 1: all-reduce the sums of this many consecutive steps as one message (default 1: one collective "
                         "per pricing call); 0 = ONE all-reduce of all K steps' sums after the last launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements (other stepper, 20-run launch, curve kernel, 2^24 reference)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--spinup-ms", type=float, default=400.0,
                    help="untimed launches before the W warm-up steps so the timed region runs at the "
                         "clock the chip holds under sustained load (DVFS ramps over the first ~100 ms)")
    return ap.parse_args()


def cpu_baseline(P, f, profile, grid_steps, seconds):
    """The oracle's OpenMP loops (same steppers, same Philox stream) on this host's cores, on bounded samples of the
    same workloads; throughput is per path so it compares directly.  `value` = the ZBC loop (the metric's kernel);
    `q1_curve` = BASELINE configs[0], the Q1 curve loop at 2^20 pairs x 1000 steps (src/1_bond_pricing.cu:60-71,110)."""
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
    out = {"value": 2 * n / dt / 1e6, "unit": "M paths/s", "cores": cores, "cores_how": how, "kind": "port",
           "sample": "%d antithetic pairs x %d executed steps, OpenMP %d threads, %.1f s (oracle/hw_oracle.c)"
                     % (n, o.steps_to(5.0), cores, dt),
           "single_thread_value": 2 * n1 / dt1 / 1e6,
           "single_thread_sample": "%d pairs, %.1f s" % (n1, dt1)}
    # configs[0]: Q1 bond_pricing, 2^20 pairs x 1000 steps x 100 maturities on the host cores -- the whole
    # configuration when it fits the time budget (it does from ~8 cores up), else a sample scaled per path
    o1 = hw_oracle.Oracle(profile, n_steps=1000)
    t0 = time.perf_counter()
    o1.zcb_curve(1 << 12, 1, n_threads=cores)
    r1 = (1 << 12) / max(time.perf_counter() - t0, 1e-6)
    nq = int(min(1 << 20, max(1 << 12, r1 * max(seconds, 4.0))))
    t0 = time.perf_counter()
    o1.zcb_curve(nq, 2, n_threads=cores)
    dq = time.perf_counter() - t0
    nq1 = max(nq // (4 * cores), 1 << 10)
    t0 = time.perf_counter()
    o1.zcb_curve(nq1, 3, n_threads=1)
    dq1 = time.perf_counter() - t0
    out["q1_curve"] = {"config": "BASELINE configs[0]: Q1 bond_pricing, 2^20 pairs x 1000 steps, antithetic, 100 maturities, CPU OpenMP loop",
                       "value": 2 * nq / dq / 1e6, "unit": "M paths/s", "cores": cores, "ms": dq * 1e3,
                       "sample": "%d antithetic pairs x 1000 steps (%s), OpenMP %d threads, %.1f s"
                                 % (nq, "the full configuration" if nq == 1 << 20 else "scaled per path", cores, dq),
                       "single_thread_value": 2 * nq1 / dq1 / 1e6, "single_thread_sample": "%d pairs, %.1f s" % (nq1, dq1),
                       "reference_metric": "throughput = (2*N_PATHS/ms)/1000, src/1_bond_pricing.cu:110"}
    return out


def pmc_traffic(config_is_default):
    """roofline.traffic: HBM bytes per launch from the PMC counters -- ONLY when they were measured on this box: the profiler passes
    run as separate processes (tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, never in-process)
    and hand their summary over through HWMC_BENCH_PMC_FILE.  Without it traffic is null and traffic_source names the newest
    committed summary (another box's number: a reference for the reader, not evidence of this run)."""
    committed = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")):
        m = re.match(r"r(\d+)_pmc_summary\.json$", os.path.basename(f))
        if m and (committed is None or int(m.group(1)) > committed[0]):
            committed = (int(m.group(1)), f)
    src = {"measured_in_this_run": False,
           "committed_reference": ("profiles/" + os.path.basename(committed[1])) if committed else None,
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (separate runs); FETCH_SIZE x2 on gfx950; "
                  "set HWMC_BENCH_PMC_FILE to a summary taken on this box (tools/profile_round.sh does)"}
    if committed:
        try:
            with open(committed[1]) as fh:
                src["committed_reference_bytes_per_launch"] = json.load(fh)["hbm_traffic_bytes_per_launch"]["corrected_bytes"]
        except (OSError, KeyError, ValueError):
            pass
    path = os.environ.get("HWMC_BENCH_PMC_FILE")
    if not path or not config_is_default:
        return None, src
    try:
        with open(path) as fh:
            d = json.load(fh)
        src.update(measured_in_this_run=True, file=path, kernel=d.get("kernel"), command=d.get("command"))
        return d["hbm_traffic_bytes_per_launch"]["corrected_bytes"], src
    except (OSError, KeyError, ValueError) as exc:
        src["error"] = "HWMC_BENCH_PMC_FILE unreadable: %s" % exc
        return None, src


class Checks:
    """The line's own correctness checks: every timed side launch looks at what it computed."""

    def __init__(self):
        self.passed, self.failed = [], []
        self.z = {}                                        # name -> distance in standard errors, for the statistical checks

    def add(self, name, ok, detail=None):
        (self.passed if ok else self.failed).append(name if ok or detail is None else "%s: %s" % (name, detail))
        return ok

    def within_se(self, name, got, want, se, n_se=4.0, slack=0.0):
        """|got - want| <= n_se standard errors (+ slack for float32 storage / discretisation)"""
        z = abs(got - want) / max(se, 1e-300)
        self.z[name.split(":")[0]] = round(z, 2)
        return self.add(name, abs(got - want) <= n_se * se + slack, "%.8g vs %.8g = %.2f SE" % (got, want, z))

    def rel(self, name, a, b, tol):
        import numpy as np
        a, b = np.asarray(a, float), np.asarray(b, float)
        worst = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.shape == b.shape and a.size else float("inf")
        return self.add(name, worst <= tol, "max rel diff %.3e > %.1e" % (worst, tol))

    def equal(self, name, a, b):
        import numpy as np
        return self.add(name, np.array_equal(np.asarray(a), np.asarray(b)), "not bit-identical")


def build_claims(line, passed, failed):
    """The compact block that ends the N = 1 line (<= 1.5 KB as JSON): every rate README.md's table quotes -- value in M paths/s,
    ms per call or launch, share of the measured VALU-issue roof --, the size table and the stress launches with their HBM
    fractions (north_star: 'as fraction of the HBM roofline'), the two vs_baseline countings and the outcome of the line's own
    checks.  Built from the finished line alone, so tests/test_bench_helpers.py can rebuild it from a committed line."""
    def dig(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    v = line.get("valu") or {}
    kiso = dig(line, "roofline", "kernel_ms_avg")
    cyc_per_ms = (v.get("achieved_cycles_per_block_isolated_launch") or 0.0) / kiso if kiso else None   # of a launch of this step's size
    bound = v.get("bound_cycles_per_block")
    vf = lambda ms: (bound / (ms * cyc_per_ms)) if (bound and cyc_per_ms and ms) else None
    rate = lambda val, ms, fr: [sig(val, 6), sig(ms), sig(fr, 4)]
    rates = {"kernel_alone": rate(line.get("kernel_Mpaths_per_sec_isolated_launch"), kiso, v.get("frac")),
             "serial": rate(dig(line, "serial", "value"), dig(line, "serial", "ms_per_step"), vf(dig(line, "serial", "ms_per_step"))),
             "two_queues": rate(line.get("value"), line.get("ms_per_step"), v.get("frac_timed_region"))}
    if "sync_call" in line:
        rates["sync_call"] = rate(line["sync_call"]["value"], line["sync_call"]["ms_per_call"], vf(line["sync_call"]["ms_per_call"]))
    for k in ("library_pipeline", "long_region"):
        if k in line:
            rates[k] = rate(line[k]["value"], line[k]["ms_per_step"], vf(line[k]["ms_per_step"]))
    if "runs20_single_launch" in line:
        r20 = line["runs20_single_launch"]
        rates["runs20_single_launch"] = rate(r20["value"], r20["ms"], vf(r20["ms"] / r20["n_runs"]))
    if "other_stepper" in line:
        rates["other_stepper"] = rate(line["other_stepper"]["value"], line["other_stepper"]["ms_per_step"], None)   # (its loop has no twin)
    for k in ("q1_curve_gpu", "q3_vega_gpu", "q3_fd_vega_gpu"):
        if k in line:
            rates[k] = rate(line[k]["value"], line[k]["kernel_ms"], dig(line[k], "valu", "frac"))
    claims = {"unit": "M paths/s | ms per call or launch | share of the measured VALU-issue roof",
              "rates_cols": ["value", "ms", "valu_frac"], "rates": rates}
    if "size_table" in line:
        claims["size_cols"] = ["pairs_log2", "value", "hbm_frac", "valu_frac"]
        claims["size_table"] = [[r["pairs_log2"], sig(r["value"], 6), sig(r.get("hbm_frac"), 3), sig(r.get("valu_frac"), 4)]
                                for r in line["size_table"]["rows"]]
    if "stress" in line:
        claims["stress_cols"] = ["ms", "value", "hbm_frac", "valu_frac"]
        claims["stress"] = {p: {k: [sig(line["stress"][p][k]["kernel_ms"]), sig(line["stress"][p][k]["value"], 6),
                                    sig(dig(line["stress"][p][k], "hbm", "frac"), 3), sig(dig(line["stress"][p][k], "valu", "frac"), 4)]
                                for k in ("zbc", "curve")} for p in ("f32", "f64")}
    claims.update(hbm_frac=sig(dig(line, "roofline", "frac"), 3), vs_baseline=line.get("vs_baseline"),
                  vs_baseline_counting=dig(line, "config", "counting"), vs_baseline_like_for_like=line.get("vs_baseline_like_for_like"),
                  checks_passed=len(passed), checks_failed=list(failed))
    return claims


def sig(x, digits=5):
    """x rounded to `digits` significant digits (None stays None): keeps `claims` short"""
    if x is None:
        return None
    x = float(x)
    if x == 0.0 or not math.isfinite(x):
        return x
    return round(x, digits - 1 - int(math.floor(math.log10(abs(x)))))


def mix_bound(live=True):
    """The measured issue bound of the float (y, W) loop's instruction mix (tools/mix_bound.hip): live from tools/bin/mix_bound
    when it is built -- call BEFORE this process initialises the GPU -- else the newest committed profiles/rNN_mix_bound.json."""
    exe = os.path.join(ROOT, "tools", "bin", "mix_bound")
    traced = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if live and not traced and os.path.exists(exe) and os.environ.get("HWMC_BENCH_NO_LIVE_BOUND") != "1":   # (never a child under a profiler)
        import subprocess
        try:
            r = subprocess.run([exe, "200000", "--json"], capture_output=True, text=True, timeout=120)
            for l in r.stdout.splitlines():
                if l.startswith("{"):
                    d = json.loads(l)
                    d["source"] = "live: tools/bin/mix_bound 200000 --json, run by this process before it touched the GPU"
                    return d
        except (OSError, ValueError, subprocess.SubprocessError):
            pass
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_mix_bound.json")):
        m = re.match(r"r(\d+)_mix_bound\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None
    try:
        with open(best[1]) as fh:
            d = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])
        d["source"] = "profiles/" + os.path.basename(best[1])
        return d
    except (OSError, ValueError):
        return None


def loop_bound(bound, name):
    """cycles per Philox block per SIMD of the dependency-free twin of loop `name` ("zbc_f32", "zbc_f64", "curve_f32_p1",
    "curve_f32_p2", "curve_f64_p2"; tools/mix_bound.hip `loops`), or None when the probe result does not carry it."""
    if not bound:
        return None
    if name == "zbc_f32":
        return bound.get("bound_cycles_per_block")
    d = (bound.get("loops") or {}).get(name)
    return d.get("bound_cycles_per_block") if d else None


def union_ms(spans):
    """Total length of the union of [start, end] intervals."""
    total, cur_a, cur_b = 0.0, None, None
    for a, b in sorted(spans):
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                total += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        total += cur_b - cur_a
    return total


def main():
    a = parse()
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)  # a default run takes well under a minute: never sit in a hang
    rank0 = int(os.environ.get("RANK", "0")) == 0
    exit_code = 0
    # the measured VALU issue bound (a child process on the GPU: before this process initialises HIP; rank 0, N = 1 only)
    bound = mix_bound(live=rank0 and a.gpus == 1) if rank0 else None
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
            # HWMC_BENCH_RCCL_PRIORITY=high: torch's RCCL stream is created with high priority, so the tiny all-reduce
            # kernels are dispatched ahead of the path kernels' pending workgroups (A/B switch)
            if os.environ.get("HWMC_BENCH_RCCL_PRIORITY", "normal") == "high":
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                dist.init_process_group("nccl", device_id=dev, pg_options=opts)
            else:
                dist.init_process_group("nccl", device_id=dev)
    cdev = "cpu" if rehearsal else dev                 # where the small control tensors of the collectives live
    # a second, host-side group (gloo) for the side measurement "the same pricing call with the sums added on the host"
    host_group = None
    if dist_on and not rehearsal and os.environ.get("HWMC_BENCH_HOST_GROUP", "1") != "0":
        try:                                           # (every rank takes the same branch: same host, same torch build)
            import datetime
            host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))
        except Exception as exc:                       # a torch without gloo, an interface gloo cannot bind: the side measurement is skipped
            print("bench.py: no host-side group (%s); multi_gpu.single_call_host_sum_ms will be null" % exc, file=sys.stderr)
            host_group = None
        # every rank must take the same branch later (the side measurement holds collectives): agree on it over the main group
        ok = torch.tensor([1.0 if host_group is not None else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            host_group = None

    import numpy as np
    import hwmc
    # ---- workload -------------------------------------------------------------------------------------------
    n_total = hwmc.parallel.pricing_workload(world, a.scaling, a.paths_log2, a.total_log2)
    offset, shard_n = hwmc.parallel.shard(n_total, rank, world)    # rank g owns global pairs [offset, offset + shard_n)
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
    K, W, R = a.steps, a.warmup, a.regions               # R = 0: decided after the first timed region (below)
    out = torch.zeros((K + W, 5), dtype=torch.float64, device=dev)
    B = max(1, a.allreduce_every) if a.allreduce_every > 0 else K

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The communicator (and the stream torch's RCCL backend runs its kernels on) must exist BEFORE the compute streams
    # are chosen: the choice below measures the workload INCLUDING the per-step all-reduce.
    if dist_on:
        warm = torch.zeros(5, dtype=torch.float64, device=cdev)
        dist.all_reduce(warm)
        fence()
    comm_stream = torch.cuda.Stream(device=dev)

    class _Done:
        def wait(self):
            return None

    def all_reduce_rows(i0, i1, on_streams):
        """One all-reduce of out[i0:i1+1] (contiguous), ordered after the compute streams that produced the rows."""
        if rehearsal:                                                   # host round trip, synchronous
            torch.cuda.synchronize()
            tmp = out[i0:i1 + 1].cpu()
            dist.all_reduce(tmp)
            out[i0:i1 + 1].copy_(tmp)
            return _Done()
        # torch's RCCL backend runs every collective on a stream of its own, which it makes wait for an event of the
        # stream that is CURRENT at the call; the caller's stream only waits for the collective in Work.wait().  So no
        # compute stream ever waits for RCCL.  One producer (the per-call default): issue it with the producer's stream
        # current -- one event hop; measured 2 % cheaper than going through a side stream (tools/labs/allreduce_cost_probe.py:
        # 12 159 plain, 11 910 this way, 11 729 through a side stream, one-rank communicator).  Several producers (a
        # message of several steps): a side stream joins them first.
        if len(on_streams) == 1:
            with torch.cuda.stream(on_streams[0]):
                return hwmc.parallel.allreduce_sums(out[i0:i1 + 1], async_op=True, force=True)
        for st in on_streams:
            comm_stream.wait_stream(st)                                 # event record + wait: the compute streams run on
        with torch.cuda.stream(comm_stream):
            return hwmc.parallel.allreduce_sums(out[i0:i1 + 1], async_op=True, force=True)

    # Explicit side streams: step i runs on streams[i % S], handed to the C ABI as a raw hipStream_t.
    # HIP multiplexes its streams onto a few hardware queues (4 here) and two streams that share a queue
    # serialise (tools/labs/queue_map_probe.py: which streams share one depends on the creation history of the whole
    # process).  So take 8 candidate streams and keep those that run the workload fastest together --
    # with the per-step all-reduce in the loop when there is one, so that the queue RCCL's stream landed on counts.
    cands = [torch.cuda.Stream(device=dev) for _ in range(8 if S > 1 else 1)]
    assert all(st.cuda_stream != 0 for st in cands)
    pairing = None
    if S > 1:
        cal_out = out                                                   # rows 0..7 (K + W >= 8 not required: modulo)
        with_ar = dist_on and not rehearsal and a.allreduce_every == 1

        def trial(chosen, with_ar=with_ar, launches=16, reps=2):
            best = float("inf")
            for _ in range(reps):
                torch.cuda.synchronize()
                tc = time.perf_counter()
                works = []
                for j in range(launches):
                    row = j % (K + W)
                    k = j % len(chosen)
                    engines[k].enqueue_ZBC(cal_out[row].data_ptr(), shard_n, 7 + j, path_offset=offset, stepper=stepper,
                                           stream=cands[chosen[k]].cuda_stream)
                    if with_ar:
                        works.append(all_reduce_rows(row, row, [cands[chosen[k]]]))
                for w in works:
                    w.wait()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - tc) * 1e3 / launches)
            return best

        t_cal = time.perf_counter()
        while (time.perf_counter() - t_cal) * 1e3 < min(a.spinup_ms, 150.0):   # clocks up before comparing anything
            trial([0], with_ar=False)                                   # (a time-bounded loop must not contain collectives)
        # Two streams: every pair of the 8 candidates is tried (28 pairs; which pairs share a hardware queue -- with each
        # other or with the stream RCCL runs on -- is not predictable); more than two: greedy from the best pair.
        # Every rank runs the same NUMBER of trials, hence of collectives; which streams win is local to the rank.
        table = {}
        pairs = {(i, j): trial([i, j]) for i in range(len(cands)) for j in range(i + 1, len(cands))}
        chosen = list(min(pairs, key=pairs.get))
        table[2] = {"%d,%d" % k: round(v, 4) for k, v in sorted(pairs.items(), key=lambda kv: kv[1])[:6]}
        while len(chosen) < S:
            scores = {j: trial(chosen + [j]) for j in range(len(cands)) if j not in chosen}
            table[len(chosen) + 1] = {str(j): round(v, 4) for j, v in scores.items()}
            chosen.append(min(scores, key=scores.get))
        streams = [cands[j] for j in chosen]
        pairing = {"candidates": len(cands), "chosen": chosen, "ms_per_step_best_pairs": table,
                   "ms_per_step_single_stream": round(trial([chosen[0]]), 4), "all_reduce_in_the_trial": with_ar}
    else:
        streams = cands
    torch.cuda.synchronize()
    torch.cuda.set_stream(streams[0])

    def enqueue(i, seed, dst, n_streams=S, which=stepper, n=None, off=None, eng_list=None, own_stream=False):
        """own_stream: the context's own stream instead of this process's torch stream (no tail-event record at the end of the
        call: include/hw_mc.h, 'Stream lifetime' -- 3 us per call when nothing else is in flight, profiles/r03_seal_cost_probe.txt)."""
        k = i % n_streams
        (eng_list or engines)[k].enqueue_ZBC(dst.data_ptr(), shard_n if n is None else n, seed,
                                             path_offset=offset if off is None else off, stepper=which,
                                             stream=None if own_stream else streams[k].cuda_stream)
        return k

    reducer = hwmc.parallel.StepReducer(a.allreduce_every, K + W,
                                        lambda i0, i1, ks: all_reduce_rows(i0, i1, [streams[j] for j in ks]))

    def step(i, seed, last=False):
        k = enqueue(i, seed, out[i])
        return reducer.after_step(i, producer=k, last=last) if dist_on else None

    def join_streams():                                                 # streams[0] waits for the others
        for st in streams[1:]:
            streams[0].wait_stream(st)

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

    # ---- the timed regions: EXACTLY K steps each, barrier + synchronize on both sides -----------------------------
    # One HIP event pair brackets the K launches of a region: recorded on streams[0] before the first launch and, after
    # streams[0] has been made to wait for the other streams, behind the last one.
    elapsed_regions, region_ms_events, region_wall_start = [], [], []
    r_done = 0
    while True:
        if R > 0 and r_done >= R:
            break
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        works = []
        fence()
        region_wall_start.append(time.time())              # (one node, one host clock: comparable across ranks -> multi_gpu.launch_skew)
        t0 = time.perf_counter()
        ev0.record(streams[0])
        for i in range(K):
            works.append(step(W + i, 1234 + 12345 * i, last=(i == K - 1)))   # seeds as 2_option:229: base + run*12345
        join_streams()
        ev1.record(streams[0])
        for w in works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize()                           # every local launch AND every all-reduce has completed, i.e.
        elapsed_regions.append(time.perf_counter() - t0)   # every rank has contributed all its K steps; MAX over ranks below
        fence()
        region_ms_events.append(ev0.elapsed_time(ev1))
        r_done += 1
        if R <= 0:
            # automatic region count: every rank must run the SAME number of regions (each holds barriers and, for N > 1,
            # collectives), so it is derived from the MAX over ranks of the first region's time
            t1 = torch.tensor([elapsed_regions[0]], dtype=torch.float64, device=cdev)
            if dist_on:
                dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            R = int(min(400, max(7, math.ceil(1.05 * TIMED_TOTAL_S / max(float(t1.item()), 1e-6)))))   # (+5 %: later regions run a little faster than the first)
    sums = out[W:].cpu().numpy()                           # the K all-reduced moment vectors of the timed steps
    if dist_on:                                            # MAX over ranks, region by region (one collective)
        t = torch.tensor(elapsed_regions, dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_regions = [float(x) for x in t.cpu()]
    elapsed = float(np.median(elapsed_regions))
    chk = Checks()                                         # the line's own correctness checks (rank 0 reports them)
    exact_price, exact_vega = hwmc.closed_form_zbc(p, P), hwmc.closed_form_vega(p, P)   # Jamshidian on the market curve; d/dsigma
    se_of = lambda r_: r_["ci95_halfwidth"] / 1.96         # standard error of one control-variate price

    def check_price(name, sums5, n_pairs, P_=P, exact=None, slack=0.0):
        r_ = hwmc.cv_price(sums5, 2 * n_pairs, float(P_[-1]))
        chk.within_se(name, r_["price"], exact_price if exact is None else exact, se_of(r_), slack=slack)
        return r_

    # ---- untimed side measurements ---------------------------------------------------------------------------------
    scratch = torch.zeros((max(K, 8), 5), dtype=torch.float64, device=dev)

    def respin(ms=60.0):
        """Host-side bookkeeping between the passes (reading events back, Python) leaves the chip idle for milliseconds and
        its clock starts to drop; every measurement pass below is preceded by a short untimed burst of the same launches."""
        ts_ = time.perf_counter()
        while (time.perf_counter() - ts_) * 1e3 < ms:
            for j in range(8):
                enqueue(j, 7, scratch[j % K])
            torch.cuda.synchronize()

    # pass 1: the same K launches in the same alternation with a hipEvent pair tight around every path kernel: the union of
    # those dispatch intervals / K is the time the path kernel occupied the chip per step of the (replayed) timed region
    respin()
    for eng in engines:
        eng.profile_enable(K)
    for i in range(K):
        enqueue(i, 1234 + 12345 * i, scratch[i])
    torch.cuda.synchronize()
    spans = []
    for eng in engines:
        st_ms, en_ms = eng.profile_spans(ref=engines[0])
        spans += list(zip(st_ms.tolist(), en_ms.tolist()))
    kernel_ms = np.concatenate([eng.profile_read(K) for eng in engines])
    assert kernel_ms.size == K and len(spans) == K
    chip_ms_per_step = union_ms(spans) / K
    # pass 2: the same K launches strictly one after another on ONE stream: isolated kernel durations ...
    respin()
    e.profile_enable(K)
    for i in range(K):
        enqueue(i, 1234 + 12345 * i, scratch[i], n_streams=1)
    torch.cuda.synchronize()
    kernel_ms_isolated = e.profile_read(K)
    for eng in engines:
        eng.profile_enable(0)                              # event pairs off: they cost queue bubbles
    # ... and the wall-clock rate without pipelining and without event pairs (median of the same number of regions)
    serial_regions, serial_caller_regions = [], []
    R_side = min(R, 15)                                    # the untimed side measurements repeat at most 15 times
    respin()
    for own in (False, True):                              # on this process's stream, then on the context's own stream
        for _ in range(R_side):
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(K):
                enqueue(i, 1234 + 12345 * i, scratch[i], n_streams=1, own_stream=own)
            torch.cuda.synchronize()
            (serial_regions if own else serial_caller_regions).append((time.perf_counter() - ts) * 1e3 / K)
    serial_ms = float(np.median(serial_regions))
    serial_caller_ms = float(np.median(serial_caller_regions))
    if world == 1:
        chk.equal("serial: one stream gives the timed region's sums bit for bit", scratch[:K].cpu().numpy(), sums)
    clock_mhz = e.profile_clock_mhz() if a.stepper == "superposed" else None
    geo = e.launch_geometry("zbc" if a.stepper == "superposed" else "zbc_ref", shard_n)

    # N > 1: one pricing call end to end, strictly serial -- launch, all-reduce, host sees the result -- i.e. the latency
    # of a single sharded pricing call (what a caller of hw_comm_zbc_moments / bin/q2 --gpus N waits for)
    single_call_ms = None
    allreduce_alone_ms = None
    if dist_on:
        fence()
        tt = []
        for i in range(min(K, 10)):
            dist.barrier()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            enqueue(0, 1234 + 12345 * i, out[0], n_streams=1)
            w = all_reduce_rows(0, 0, [streams[0]])
            w.wait()
            torch.cuda.synchronize()
            tt.append((time.perf_counter() - tc) * 1e3)
        tsc = torch.tensor([float(np.median(tt))], dtype=torch.float64, device=cdev)
        dist.all_reduce(tsc, op=dist.ReduceOp.MAX)
        single_call_ms = float(tsc.item())
        tt = []
        for _ in range(10):                                # the collective alone (5 doubles), host-synchronous
            torch.cuda.synchronize()
            tc = time.perf_counter()
            w = all_reduce_rows(1, 1, [])
            w.wait()
            torch.cuda.synchronize()
            tt.append((time.perf_counter() - tc) * 1e3)
        allreduce_alone_ms = float(np.median(tt))
        fence()
    per_rank_kernel_ms = None
    seen = skew = compute_only_ms = single_call_host_ms = None
    if dist_on:
        mine = torch.tensor([float(np.mean(kernel_ms_isolated))], dtype=torch.float64, device=cdev)
        allk = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        per_rank_kernel_ms = [float(x.item()) for x in allk]
        # what an N-GPU line needs to check itself (hwmc.parallel.check_scaling_lines): who took part, how far apart the
        # ranks left the barrier, and what the collectives cost on top of the same K steps without them
        seen = hwmc.parallel.world_size_seen(cdev)
        skew = hwmc.parallel.launch_skew(region_wall_start[:R_side], cdev)
        co = []
        for _ in range(R_side):                            # the timed region again, same alternation, NO all-reduce
            fence()
            tc = time.perf_counter()
            for i in range(K):
                enqueue(i, 1234 + 12345 * i, scratch[i])
            torch.cuda.synchronize()
            co.append((time.perf_counter() - tc) * 1e3 / K)
            fence()
        tco = torch.tensor(co, dtype=torch.float64, device=cdev)
        dist.all_reduce(tco, op=dist.ReduceOp.MAX)
        compute_only_ms = float(np.median(tco.cpu().numpy()))
        # one sharded pricing call with the five sums added on the HOST (device -> host copy, gloo all-reduce) instead of RCCL
        hg = None if rehearsal else host_group
        if rehearsal or hg is not None:
            tt = []
            for i in range(min(K, 10)):
                dist.barrier()
                torch.cuda.synchronize()
                tc = time.perf_counter()
                enqueue(0, 1234 + 12345 * i, out[0], n_streams=1)
                torch.cuda.synchronize()
                tmp = out[0].cpu()
                dist.all_reduce(tmp, group=hg)
                tt.append((time.perf_counter() - tc) * 1e3)
            th = torch.tensor([float(np.median(tt))], dtype=torch.float64, device=cdev)
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            single_call_host_ms = float(th.item())
        fence()

    extras = {}
    if world == 1 and not a.no_extras and S > 1:
        # The same alternation over 10 K steps in ONE region: what the two edges of a K-step region cost.  After a
        # synchronisation both queues start together and the launches need ~10 steps to settle into the staggered rhythm in
        # which one launch's drain is filled by the other queue (tools/labs/region_edge_probe.py: T = ~60 us + K x steady).
        respin()
        k_long = 10 * K
        torch.cuda.synchronize()
        tl = time.perf_counter()
        for i in range(k_long):
            enqueue(i, 1234 + 12345 * i, scratch[i % K])
        torch.cuda.synchronize()
        long_ms = (time.perf_counter() - tl) * 1e3 / k_long
        check_price("long_region: price of its last step on row 0 within 4 SE", scratch[0].cpu().numpy(), shard_n)
        extras["long_region"] = {"steps": k_long, "ms_per_step": long_ms, "value": (2 * shard_n / long_ms) / 1000.0, "unit": "M paths/s",
                                 "note": "untimed side measurement: 10 K steps between two synchronisations instead of K; the difference to "
                                         "`value` is the cost of starting and draining the two-queue pipeline once per K steps"}
    if world == 1 and not a.no_extras and S > 1 and a.stepper == "superposed":
        # The same K pricing calls through the LIBRARY's own pipe (hw_pipe_*: the C ABI's form of what the timed region does
        # by hand with two contexts and a stream search): K requests in, K x 5 sums out on the host, one synchronisation.
        with hwmc.HullWhitePipeline(p, device=local_rank, queues=S) as pipe:
            pipe.load_market_data(P, f)
            tuned_ms = pipe.tune(shard_n)
            seeds = [1234 + 12345 * i for i in range(K)]
            tp = time.perf_counter()
            while (time.perf_counter() - tp) * 1e3 < 60.0:
                pipe.simulate_ZBC_batch(shard_n, seeds, path_offset=offset)
            got, ms_list = None, []
            for _ in range(R_side):
                got, ms = pipe.simulate_ZBC_batch(shard_n, seeds, path_offset=offset, return_ms=True)
                ms_list.append(ms / K)
            chk.equal("library_pipeline: hw_pipe_zbc_moments returns the timed region's sums bit for bit", got, sums)
            pm = float(np.median(ms_list))
            extras["library_pipeline"] = {"api": "hw_pipe_zbc_moments", "queues": S, "steps": K, "ms_per_step": pm,
                                          "value": (2 * shard_n / pm) / 1000.0, "unit": "M paths/s", "tuned_ms_per_step": tuned_ms,
                                          "note": "K requests through the C ABI's pipe, host wall time inside the call (enqueue, one D2H "
                                                  "copy of the K x 5 sums, one synchronisation), median of %d batches; sums identical to the timed region's" % R_side}
    if world == 1 and not a.no_extras:
        # the OTHER stepper on the same workload: the reference's per-step arithmetic verbatim vs the superposed recursion
        other = hwmc.HW_STEPPER_REFERENCE if a.stepper == "superposed" else hwmc.HW_STEPPER_SUPERPOSED
        if not (other == hwmc.HW_STEPPER_REFERENCE and a.precision != "f32"):
            tw = time.perf_counter()                       # a changed kernel runs its first tens of launches slow (power
            while (time.perf_counter() - tw) * 1e3 < 80.0:  # management settles anew): warm it like the headline kernel
                for i in range(8):
                    enqueue(i, 5, scratch[i % K], which=other)
                torch.cuda.synchronize()
            ta = time.perf_counter()
            for i in range(K):
                enqueue(i, 1234 + 12345 * i, scratch[i], which=other)
            torch.cuda.synchronize()
            alt_ms = (time.perf_counter() - ta) * 1e3 / K
            chk.rel("other_stepper: the two steppers agree on the same draws to 1e-5", scratch[:K].cpu().numpy(), sums, 1e-5)
            extras["other_stepper"] = {"stepper": "reference" if other == hwmc.HW_STEPPER_REFERENCE else "superposed",
                                       "ms_per_step": alt_ms, "value": (2 * shard_n / alt_ms) / 1000.0, "unit": "M paths/s"}
        if a.stepper == "superposed":
            # the 20-run validation as ONE launch (hw_zbc_moments_runs): 20 x n pairs in a single grid, no pipelining needed
            # to cover the launch overhead and the drain -- the non-pipelined way to run the reference's 20-run harness
            respin()                                       # (the reference-stepper kernel ran last)
            ms20 = []
            for _ in range(5):
                rows20, ms = e.simulate_ZBC_runs(shard_n, 1234, n_runs=20, seed_stride=12345, return_ms=True)
                ms20.append(ms)
            m20 = float(np.median(ms20))
            chk.equal("runs20_single_launch: per-run sums bit-identical to the separate launches", rows20[:min(K, 20)], sums[:min(K, 20)])
            extras["runs20_single_launch"] = {"n_runs": 20, "pairs_per_run": shard_n, "ms": m20, "value": (20 * 2 * shard_n / m20) / 1000.0,
                                              "unit": "M paths/s", "note": "path kernel + column pass of hw_zbc_moments_runs, hipEvent pair, median of 5"}
        def isolated_ms(eng, fn, reps, pair=True):
            """hipEvent pair tight around the path kernel of `reps` launches; pair: each timed launch directly follows an
            untimed one (a launch that follows the idle time of reading events back runs slow)."""
            out_ms = []
            for _ in range(reps):
                if pair:
                    fn()
                eng.profile_enable(1)
                fn()
                out_ms.append(float(eng.profile_read(1)[0]))
            return out_ms

        def clock_of(eng):
            """in-kernel clock of the engine's last profiled launch; kernels without the probe: the headline kernel's clock"""
            try:
                return eng.profile_clock_mhz()
            except hwmc.HwError:
                return clock_mhz or e.device_info()["clock_mhz"]

        def valu_of(ms, n_units, steps, mhz_, loop):
            """bound / achieved for one launch alone on the chip: achieved = launch time x clock x SIMDs / (waves x Philox
            blocks per path); bound = the measured dependency-free rate of that loop's own instruction stream."""
            info_ = e.device_info()
            cyc = ms * 1e-3 * mhz_ * 1e6 * info_["compute_units"] * 4 / (((n_units + 63) // 64) * (steps / 4.0))
            b = loop_bound(bound, loop)
            return {"loop": loop, "achieved_cycles_per_block_isolated_launch": cyc, "bound_cycles_per_block": b,
                    "frac": (b / cyc) if b else None, "clock_mhz_in_kernel": mhz_}

        # the Q1 curve kernel at BASELINE configs[0]'s size (2^20 pairs x 1000 steps x 100 maturities), beside cpu_baseline.q1_curve
        pc = hwmc.default_params(a.profile, n_steps=1000, precision=a.precision)
        with hwmc.HullWhiteEngine(pc, device=local_rank) as ec:
            tw = time.perf_counter()                       # (80 ms are not enough here: inside this process the curve kernel's
            while (time.perf_counter() - tw) * 1e3 < 500.0:  #  launches were still getting faster after 12 timed pairs, 0.455 -> 0.419 ms)
                ec.simulate_zcb(1 << 20, 1)
            msq = []
            for j in range(12):
                ec.simulate_zcb(1 << 20, 1234 + j)          # back-to-back pair: the second launch is the timed one
                ec.profile_enable(1)
                psum = ec.simulate_zcb(1 << 20, 1234 + j)
                msq.append(float(ec.profile_read(1)[0]))
            mq = float(np.median(msq))
            Pc, _ = hwmc.analytic_curve(pc)

            def check_curve(name, psum_, n_, pp_, Pex):
                """path count exact; every P(0,T) within 4 SE (+ float32 storage of the exact curve) of the closed form"""
                sec = hwmc.curve_standard_errors(pp_, Pex, n_)
                z = np.abs(psum_[1:] / (2.0 * n_) - np.asarray(Pex, np.float64)[1:]) / (sec[1:] + 5e-8)
                chk.z[name.split(":")[0]] = round(float(z.max()), 2)
                chk.add(name, bool(psum_[0] == 2.0 * n_ and z.max() <= 4.0), "P_sum[0] = %r, worst maturity %d at %.2f SE" % (psum_[0], int(z.argmax()) + 1, z.max()))
            check_curve("q1_curve_gpu: path count exact, every P(0,T) within 4 SE of the exact curve", psum, 1 << 20, pc, Pc)
            if os.environ.get("HWMC_BENCH_TRACE"):
                print("bench: q1 curve launches ms", ["%.4f" % x for x in msq], file=sys.stderr, flush=True)
            extras["q1_curve_gpu"] = {"pairs": 1 << 20, "grid_steps": 1000, "maturities": 100, "kernel_ms": mq,
                                      "value": (2 * (1 << 20) / mq) / 1000.0, "unit": "M paths/s",
                                      "vs_v100_391": ((2 * (1 << 20) / mq) / 1000.0) / V100_PUBLISHED_Q1_MPATHS,
                                      # the curve loop's OWN measured issue bound (Philox + Box-Muller + two half-block steps + the
                                      # maturity events of a 16-maturity period, dependency-free: tools/gen_mix_bound.py CURVE*)
                                      "valu": valu_of(mq, 1 << 20, 1000, clock_of(ec), "curve_%s_p1" % a.precision)}
        # the Q3 pathwise-Vega kernel at BASELINE configs[2]'s size (2^20 single paths, 500 executed steps), metric exactly
        # as src/3_sensitivity_analysis.cu:266: (N_PATHS / ms) / 1000 with the event pair around the kernel
        respin()
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < 80.0:
            e.simulate_sensitivity(1 << 20, 1)
        msv = []
        for j in range(5):
            e.simulate_sensitivity(1 << 20, 4242 + j)       # back-to-back pair: the second launch is the timed one
            e.profile_enable(1)
            vsum = e.simulate_sensitivity(1 << 20, 4242 + j)
            msv.append(float(e.profile_read(1)[0]))
        mv = float(np.median(msv))
        # its standard error from eight batch means of 2^17 paths (one more launch, untimed), then the timed launch's own result
        bm = e.simulate_sensitivity_runs(1 << 17, 99, n_runs=8) / float(1 << 17)
        se_vega = float(np.std(bm, ddof=1)) / math.sqrt(8.0)
        chk.within_se("q3_vega_gpu: pathwise Vega within 4 SE of d(closed form)/d(sigma)", vsum / float(1 << 20), exact_vega, se_vega)
        extras["q3_vega_gpu"] = {"paths": 1 << 20, "grid_steps": a.grid_steps, "kernel_ms": mv, "value": ((1 << 20) / mv) / 1000.0,
                                 "kernel_ms_for_valu": mv,
                                 "unit": "M paths/s", "vs_v100_509": (((1 << 20) / mv) / 1000.0) / V100_PUBLISHED_Q3_MPATHS,
                                 "reference_metric": "throughput = (N_PATHS/ms)/1000, src/3_sensitivity_analysis.cu:266 (single paths, not pairs)",
                                 "note": "pathwise Vega kernel alone on the chip, hipEvent pair around the path kernel, median of 5"}
        if a.stepper == "superposed":
            steps_s1 = int(round(5.0 * a.grid_steps / 10.0))
            # (a) the finite-difference half of BASELINE configs[2]: sigma, sigma -+ eps priced from ONE pass over the same draws
            #     (hw_zbc_scenarios) = the three launches + two state restores of run_finite_difference, 3_sens:400-446
            respin()
            sg0, eps = p.sigma, 1e-3
            fd_fn = lambda sd=7: e.simulate_ZBC_scenarios(1 << 20, sd, [sg0, sg0 - eps, sg0 + eps],
                                                         [hwmc.HW_DRIFT_BASE, hwmc.HW_DRIFT_SHIFTED, hwmc.HW_DRIFT_SHIFTED])
            tw = time.perf_counter()
            while (time.perf_counter() - tw) * 1e3 < 80.0:
                fd_fn()
            ms_fd = float(np.median(isolated_ms(e, fd_fn, 5)))
            mhz_fd = clock_of(e)
            sc = fd_fn(1234)
            pr = [hwmc.cv_price(x, 2 << 20, float(P[-1]))["price"] for x in sc]
            check_price("q3_fd_vega_gpu: base-scenario price within 4 SE", sc[0], 1 << 20)
            chk.within_se("q3_fd_vega_gpu: finite-difference Vega within 4 SE of d(closed form)/d(sigma)", (pr[2] - pr[1]) / (2.0 * eps), exact_vega, se_vega)
            extras["q3_fd_vega_gpu"] = {
                "pairs": 1 << 20, "scenarios": 3, "epsilon": eps, "kernel_ms": ms_fd,
                "value": (2 * (1 << 20) / ms_fd) / 1000.0, "unit": "M paths/s (per scenario: all three come from one pass)",
                "scenario_paths_per_sec_M": (3 * 2 * (1 << 20) / ms_fd) / 1000.0,
                "vega_fd": (pr[2] - pr[1]) / (2.0 * eps), "price": pr[0],
                "over_single_scenario_launch": ms_fd / float(np.mean(kernel_ms_isolated)) if shard_n == 1 << 20 else None,
                "valu": valu_of(ms_fd, 1 << 20, steps_s1, mhz_fd, "zbc_" + a.precision),
                "reference": "run_finite_difference, src/3_sensitivity_analysis.cu:400-446 (3 launches of simulate_ZBC_control_variate + 2 state restores)",
                "note": "three-scenario path kernel alone on the chip, hipEvent pair, median of 5; valu.frac against the ONE-scenario loop's "
                        "bound: the loop is the same, the two extra payoff epilogues are what the difference pays for"}
            # (b) kernel alone on the chip against batch size (north_star: 'paths/sec on 2^20-2^26 paths x 1000 steps')
            table = []
            for lg in (20, 22, 24, 26):
                n_ = 1 << lg
                last_ = {}
                fn_ = lambda n_=n_: last_.__setitem__("sums", e.simulate_ZBC_control_variate(n_, 11))
                fn_()
                ms_ = float(np.median(isolated_ms(e, fn_, 5 if lg < 26 else 3, pair=lg < 24)))
                check_price("size_table 2^%d: price within 4 SE" % lg, last_["sums"], n_)
                bytes_ = e.launch_geometry("zbc", n_)["partial_bytes"] + 40
                row = {"pairs_log2": lg, "kernel_ms": ms_, "value": (2 * n_ / ms_) / 1000.0,
                       "hbm_frac": bytes_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": bytes_,
                       "valu_frac": valu_of(ms_, n_, steps_s1, clock_of(e), "zbc_" + a.precision)["frac"]}
                table.append(row)
            extras["size_table"] = {"grid_steps": a.grid_steps, "steps_executed": steps_s1, "unit": "M paths/s", "rows": table,
                                    "note": "one launch at a time, hipEvent pair around the path kernel, median of 5 (3 at 2^26)"}
            # (c) host-to-host time of a synchronous hw_zbc_moments call (kernel_ms = NULL: no event pair, polled completion),
            #     calls strictly one after another from this thread through ctypes with a prebuilt request
            import ctypes as C_
            req = e._request(1 << 20, 1234, 0, 5.0, 10.0, None, 0.0, hwmc.HW_DRIFT_BASE, hwmc.HW_STEPPER_SUPERPOSED)
            out5 = (C_.c_double * 5)()
            call = e.lib.hw_zbc_moments
            ctx_, rq_ = e._ctx, C_.byref(req)
            respin()
            for _ in range(10):
                call(ctx_, rq_, out5, None)
            tcall = []
            for k_ in range(20):
                req.seed = 1234 + 12345 * k_
                tq = time.perf_counter()
                rc_ = call(ctx_, rq_, out5, None)
                tcall.append((time.perf_counter() - tq) * 1e3)
                assert rc_ == 0
            sc_ms = float(np.median(tcall))
            if K >= 20 and shard_n == 1 << 20:                # the 20th call priced the 20th timed step's seed
                chk.equal("sync_call: the last call's sums equal the timed region's for the same seed", np.array(out5[:]), sums[19])
            else:
                check_price("sync_call: price of the last call within 4 SE", np.array(out5[:]), 1 << 20)
            extras["sync_call"] = {"api": "hw_zbc_moments(ctx, req, out, NULL)", "pairs": 1 << 20, "calls": 20, "ms_per_call": sc_ms,
                                   "ms_min": float(np.min(tcall)), "value": (2 * (1 << 20) / sc_ms) / 1000.0,
                                   "vs_baseline": (2 * (1 << 20) / sc_ms) / 1000.0 / V100_PUBLISHED_MPATHS, "unit": "M paths/s",
                                   "note": "host wall clock around each call (perf_counter, ctypes with a prebuilt request: ~1-2 us of binding "
                                           "overhead included), 20 calls one after another, median; the result is on the host when the call returns"}
            # (d) BASELINE configs[4] on ONE GPU: 2^26 pairs x 4000-step grid, float and double engine, ZBC stepper and curve kernel
            if not os.environ.get("HWMC_BENCH_NO_STRESS"):
                stress = {"config": "BASELINE configs[4] on one GPU: 2^26 pairs x 4000-step grid (2000 executed steps to S1 = 5 for the ZBC "
                                    "stepper, 4000 for the curve), float32 vs float64 stepper",
                          "pairs": 1 << 26, "grid_steps": 4000, "launches_timed": 3}
                nS = 1 << 26
                stress_out = {}
                for prec in ("f32", "f64"):
                    ps = hwmc.default_params(a.profile, n_steps=4000, precision=prec)
                    with hwmc.HullWhiteEngine(ps, device=local_rank) as es:
                        Ps, fs = hwmc.analytic_curve(ps)
                        es.load_market_data(Ps, fs)
                        ent = {}
                        exact_s = hwmc.closed_form_zbc(ps, Ps)
                        for kname, fn_, steps_, geo_name in (
                                ("zbc", lambda: es.simulate_ZBC_control_variate(nS, 1234), 2000, "zbc"),
                                ("curve", lambda: es.simulate_zcb(nS, 1234), 4000, "zcb_blocked")):
                            fn_()                                                   # one untimed launch (43..100 ms: the clocks settle)
                            es.profile_enable(3)
                            for _ in range(3):
                                res_ = fn_()
                            msl = [float(x) for x in es.profile_read(3)]
                            stress_out[(prec, kname)] = res_
                            # the timed launch's own output: closed form within 4 SE, and the committed one-GPU constant (1e-13)
                            if kname == "zbc":
                                check_price("stress.%s.zbc: price within 4 SE of the closed form" % prec, res_, nS, P_=Ps, exact=exact_s, slack=2e-6)
                            else:
                                check_curve("stress.%s.curve: path count exact, every P(0,T) within 4 SE" % prec, res_, nS, ps, Ps)
                            key_ = hwmc.parallel.reference_key("zbc" if kname == "zbc" else "zcb", a.profile, prec,
                                                               "superposed" if kname == "zbc" else "blocked", 4000, 26, 1234)
                            if key_ in hwmc.parallel.load_reference_sums():
                                chk.rel("stress.%s.%s: equals the committed one-GPU sums (1e-13)" % (prec, kname), res_, hwmc.parallel.load_reference_sums()[key_], 1e-13)
                            ms_ = float(np.median(msl))
                            mhz_ = clock_of(es)
                            g_ = es.launch_geometry(geo_name, nS)
                            bytes_ = g_["partial_bytes"] + (5 if kname == "zbc" else ps.n_mat) * 8
                            loop = ("zbc_" + prec) if kname == "zbc" else ("curve_%s_p2" % prec)
                            ent[kname] = {"kernel_ms": ms_, "kernel_ms_all": msl, "value": (2 * nS / ms_) / 1000.0, "unit": "M paths/s",
                                          "thread_steps_per_sec": nS * steps_ / (ms_ * 1e-3),
                                          "hbm": {"algorithmic_bytes_per_launch": bytes_, "achieved_GBps": bytes_ / (ms_ * 1e-3) / 1e9,
                                                  "frac": bytes_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                                          "valu": valu_of(ms_, nS, steps_, mhz_, loop), "launch_geometry": g_}
                        stress[prec] = ent
                stress["f64_over_f32"] = {k: stress["f64"][k]["kernel_ms"] / stress["f32"][k]["kernel_ms"] for k in ("zbc", "curve")}
                chk.rel("stress.zbc: float and double stepper agree on the same draws to 1e-5", stress_out[("f32", "zbc")], stress_out[("f64", "zbc")], 1e-5)
                chk.rel("stress.curve: float and double agree on every maturity to 1e-5", stress_out[("f32", "curve")], stress_out[("f64", "curve")], 1e-5)
                extras["stress"] = stress
                respin()
        # strong-scaling reference: the N > 1 workload (2^total_log2 pairs per step) on this ONE GPU, same pipelining
        nref = 1 << a.total_log2
        if a.scaling == "strong" and nref != shard_n and a.stepper == "superposed":
            kk = 6
            for i in range(12):                              # ~35 ms of the large launches first
                enqueue(i, 5, scratch[i % 8], n=nref, off=0)
            torch.cuda.synchronize()
            tr = time.perf_counter()
            for i in range(kk):
                enqueue(i, 1234 + 12345 * i, scratch[i], n=nref, off=0)
            torch.cuda.synchronize()
            ref_ms = (time.perf_counter() - tr) * 1e3 / kk
            tr = time.perf_counter()
            for i in range(kk):
                enqueue(i, 1234 + 12345 * i, scratch[i], n=nref, off=0, n_streams=1)
            torch.cuda.synchronize()
            ref_serial_ms = (time.perf_counter() - tr) * 1e3 / kk
            ref_sums0 = [float(x) for x in scratch[0].cpu().numpy()]           # seed 1234 = the first timed step of an N-GPU run
            ref_kernel_ms = float(np.median(isolated_ms(e, lambda: e.simulate_ZBC_control_variate(nref, 1234), 3, pair=False)))
            check_price("strong_scaling_reference: price of the 2^%d-pair call within 4 SE" % a.total_log2, ref_sums0, nref)
            key_ = hwmc.parallel.reference_key("zbc", a.profile, a.precision, a.stepper, a.grid_steps, a.total_log2, 1234)
            if key_ in hwmc.parallel.load_reference_sums():
                chk.rel("strong_scaling_reference: equals the committed one-GPU sums (1e-13)", ref_sums0, hwmc.parallel.load_reference_sums()[key_], 1e-13)
            extras["strong_scaling_reference"] = {
                "pairs_per_step": nref, "n_gpus": 1, "ms_per_step": ref_ms, "value": (2 * nref / ref_ms) / 1000.0,
                "serial_ms_per_step": ref_serial_ms, "kernel_ms": ref_kernel_ms, "unit": "M paths/s",
                "sums_step0": ref_sums0,
                "sums_step0_note": "the five sums of the pricing call with seed 1234 over all 2^%d pairs: an N-GPU line's sums_step0 must equal "
                                   "them to 1e-13 (hwmc.parallel.check_scaling_lines)" % a.total_log2,
                "note": "the workload bench.py --gpus N > 1 shards (BASELINE configs[3]) run whole on one GPU: divide an N-GPU line's "
                        "ms_per_step into this one's for the strong-scaling factor"}

    if rank == 0:
        n_eff = 2 * n_total
        prices = [hwmc.cv_price(s, n_eff, float(P[-1])) for s in sums]
        st = hwmc.run_statistics([r["price"] for r in prices]) if K > 1 else {"mean": prices[0]["price"], "ci95_width": None}
        ms_per_step = elapsed * 1e3 / K
        value = (n_eff / ms_per_step) / 1000.0
        vals = sorted((n_eff / (x * 1e3 / K)) / 1000.0 for x in elapsed_regions)
        steps_exec = int(round(5.0 * a.grid_steps / 10.0))
        info = e.device_info()
        # Path-kernel durations.  kiso: hipEvent pairs tight around each of K launches issued strictly one after another =
        # the kernel alone on the chip; rocprofv3's kernel-trace average for this command run with --streams 1 reports the
        # same quantity (profiles/rNN_bench_streams1_kernel_stats.csv) and roofline.achieved uses it.
        kavg = float(np.mean(kernel_ms))
        kiso = float(np.mean(kernel_ms_isolated))
        region_per_step = float(np.median(region_ms_events)) / K
        algo_bytes = geo["partial_bytes"] + 5 * 8         # partial rows written + the 5 sums (stateless Philox: no state traffic)
        achieved = algo_bytes / (kiso * 1e-3) / 1e9
        default_cfg = (a.stepper == "superposed" and shard_n == 1 << 20 and a.grid_steps == 1000 and a.precision == "f32" and not a.block)
        traffic, traffic_src = pmc_traffic(default_cfg)
        # the K timed steps ARE the 20-run validation: their mean price against the closed form, with their own standard error
        # (the standard error of the mean from the runs' own normal-theory errors: the spread of K = 3 prices would be no estimate)
        se_mean = math.sqrt(sum(se_of(r_) ** 2 for r_ in prices)) / K
        chk.within_se("timed region: mean price of the K runs within 4 SE of the closed form", st["mean"], exact_price, se_mean, slack=1e-6)
        key0 = hwmc.parallel.reference_key("zbc", a.profile, a.precision, a.stepper, a.grid_steps, max(n_total.bit_length() - 1, 0), 1234)
        if n_total == 1 << max(n_total.bit_length() - 1, 0) and a.scaling == "strong" and key0 in hwmc.parallel.load_reference_sums():
            chk.rel("timed region: sums of its first step equal the committed one-GPU sums (1e-13)", sums[0], hwmc.parallel.load_reference_sums()[key0], 1e-13)
        vm = VALU_MODEL["superposed_f64" if (a.stepper == "superposed" and a.precision == "f64") else a.stepper]
        cyc_per_block = 2 * vm["full"] + 4 * vm["half"] + 8 * vm["quarter"]
        waves = (shard_n + 63) // 64
        simd_cycles = waves * (steps_exec / 4.0) * cyc_per_block
        simd_hz = info["compute_units"] * 4 * info["clock_mhz"] * 1e6
        nominal_frac = simd_cycles / (simd_hz * region_per_step * 1e-3)     # nominal 2/4/8 model at the NOMINAL clock: not a bound
        nominal_frac_iso = simd_cycles / (simd_hz * kiso * 1e-3)
        # achieved shader-clock cycles per Philox block (4 steps of one pair) per SIMD, at the clock the chip HELD (in-kernel
        # s_memtime / s_memrealtime probe), whole launch: prologue, epilogue, drain and the column pass included
        mhz = clock_mhz or info["clock_mhz"]
        wave_blocks = waves * (steps_exec / 4.0)            # a "block" = one Philox block (4 steps) of the 64 pairs of a wave
        cyc_of = lambda ms: ms * 1e-3 * mhz * 1e6 * info["compute_units"] * 4 / wave_blocks
        achieved_cyc, achieved_cyc_iso = cyc_of(region_per_step), cyc_of(kiso)
        use_bound = bound if (bound and a.stepper == "superposed") else None
        bcyc = loop_bound(use_bound, "zbc_" + a.precision)           # the float or the double loop's own twin
        f64_twin = ((use_bound or {}).get("loops") or {}).get("zbc_f64") if a.precision == "f64" else None
        if world == 1:
            workload = ("Q2b option_pricing (BASELINE configs[1]): 2^%d antithetic pairs x %d-step grid (%d executed to S1=5), "
                        "optimal-beta control variate, 1 x MI355X" % (a.paths_log2, a.grid_steps, steps_exec))
        elif a.scaling == "strong":
            workload = ("Q2b option_pricing sharded (BASELINE configs[3]): 2^%d antithetic pairs per pricing call x %d-step grid (%d executed), "
                        "sharded by global pair id over %d x MI355X, RCCL all-reduce of the 5 moment sums" % (a.total_log2, a.grid_steps, steps_exec, world))
        else:
            workload = ("Q2b option_pricing, weak scaling: 2^%d antithetic pairs per GPU x %d-step grid (%d executed), %d x MI355X, "
                        "RCCL all-reduce of the 5 moment sums" % (a.paths_log2, a.grid_steps, steps_exec, world))
        line = {
            "metric": "M paths/sec (ZBC option, 1000 steps)", "value": round(value, 2), "unit": "M paths/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": round(value / V100_PUBLISHED_MPATHS, 3),
            "vs_baseline_counting": ("`value` = K independent pricing calls with %d launches in flight per GPU, over the V100's one launch between two "
                                     "events (2_option_pricing.cu:131-144); like for like -- this kernel alone on the chip, the same event-pair "
                                     "timing -- is vs_baseline_like_for_like" % S),
            "vs_baseline_like_for_like": round((2 * shard_n / kiso) / 1000.0 / V100_PUBLISHED_MPATHS, 3) if world == 1 else None,
            "baseline": {"value": V100_PUBLISHED_MPATHS, "unit": "M paths/s", "hardware": "1x Tesla V100",
                         "source": "reference README.md:44,162 (single un-warmed launch)"},
            "dtype": a.precision, "data": "synthetic",
            "timed_region_s": round(float(sum(elapsed_regions)), 4),
            "value_spread": {"timed_regions": R, "steps_per_region": K, "min": round(vals[0], 2), "p10": round(vals[len(vals) // 10], 2),
                             "median": round(value, 2), "p90": round(vals[(9 * len(vals)) // 10], 2), "max": round(vals[-1], 2),
                             "region_ms_min_median_max": [round(float(x) * 1e3, 4) for x in (min(elapsed_regions), elapsed, max(elapsed_regions))],
                             "note": "`value` is the median region; every region times exactly K steps between barrier + synchronize; "
                                     "HWMC_BENCH_REGIONS_FILE=path writes every region's time there (they left the line in round 5)"},
            "config": {"workload": workload,
                       "counting": ("%d queues in flight" % S) if S > 1 else "strictly serial",
                       "pairs_per_step": n_total, "pairs_per_gpu": shard_n, "effective_paths": n_eff, "grid_steps": a.grid_steps,
                       "steps_executed": steps_exec, "stepper": a.stepper, "theta_profile": a.profile,
                       "rng": "Philox4x32-10 (rocRAND layout), stateless, subsequence = global pair id",
                       "parallelism": ("paths sharded by global pair id; the 5 f64 moment sums all-reduced over RCCL, %s"
                                       % ("one collective per pricing call" if B == 1 else "%d steps per message" % B))
                                      if world > 1 else "single GPU",
                       "allreduce_every": a.allreduce_every if dist_on else None,
                       "streams_per_gpu": S, "stream_selection": pairing,
                       "pipelining": ("the K independent steps alternate over %d contexts/streams per GPU; every step is one full launch" % S)
                                     if S > 1 else "none: steps run strictly one after another",
                       "n1_note": "N = 1 runs BASELINE configs[1] (2^20 pairs); N > 1 runs configs[3] (2^%d pairs in total, strong scaling): "
                                  "see strong_scaling_reference for the same total on one GPU" % a.total_log2},
            "thread_steps_per_sec": n_total * steps_exec / (ms_per_step * 1e-3),
            "price_mean": st["mean"], "ci95_width": st["ci95_width"], "ci95_halfwidth_single_run": prices[0]["ci95_halfwidth"],
            "beta_mean": float(np.mean([r["beta"] for r in prices])),
            "sums_step0": [float(x) for x in sums[0]],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "zbc_%s_kernel" % a.stepper, "kernel_ms_avg": kiso, "kernel_ms_min": float(np.min(kernel_ms_isolated)),
                         "kernel_ms_avg_basis": "K launches back to back on one stream, hipEvent pair around each (matches rocprofv3 --stats of this command with --streams 1)",
                         "kernel_ms_avg_overlapped_replay": kavg,
                         "kernel_ms_per_step_in_timed_region": region_per_step,
                         "kernel_ms_per_step_basis": "ONE hipEvent pair around the K launches of the timed region (recorded before the first launch and, "
                                                     "once streams[0] waits for the other streams, behind the last), / K, median region: the chip time a step "
                                                     "took inside the timed region (path kernel + column pass); it must not exceed ms_per_step",
                         "timed_region_ms_per_step_hip_events": region_per_step,
                         "kernel_ms_per_step_replay_with_event_pairs": chip_ms_per_step,
                         "kernel_ms_per_step_replay_basis": "union of the K path-kernel dispatch intervals (a hipEvent pair around EVERY launch, common clock) of an "
                                                            "untimed replay of the same alternation, / K; the event pairs themselves cost queue bubbles, so this "
                                                            "replay runs slower than the timed region",
                         "algorithmic_bytes_per_launch": algo_bytes, "launch_geometry": geo,
                         # the binding roof's own fractions as scalars of this block too (the full account is `valu`)
                         "valu_issue_bound_cycles_per_block": bcyc, "valu_issue_frac_isolated_launch": (bcyc / achieved_cyc_iso) if bcyc else None,
                         "valu_issue_frac_timed_region": (bcyc / achieved_cyc) if bcyc else None,
                         "note": "stateless Philox: the path kernel has ~0 algorithmic HBM traffic; the binding roof is VALU issue (see valu)"},
            "valu": {"bound": "valu-issue",
                     "how": "bound = measured dependency-free rate of the production ordering: shader-clock cycles per Philox block per SIMD of the "
                            "production loop's own instruction stream with its data dependencies removed, fastest of the orderings and occupancies "
                            "tried (tools/mix_bound.hip, generated from the gfx950 listing by tools/gen_mix_bound.py) -- a loop-overhead reference, "
                            "not a proven floor (that is exclusive_classes_floor); achieved = launch time x in-kernel clock x SIMDs / pair-blocks, "
                            "everything a launch does included; frac = bound / achieved",
                     "instr_per_4_steps": vm,
                     "bound_cycles_per_block": bcyc,
                     "bound_source": use_bound["source"] if use_bound else None,
                     "bound_at": (f64_twin or use_bound).get("bound_at") if use_bound else None,
                     "exclusive_classes_floor": ({"nominal_4_4_8": use_bound["exclusive_classes_nominal_floor"],
                                                  "measured": use_bound["exclusive_classes_measured"]} if use_bound and a.precision == "f32" else None),
                     "production_loop_in_the_harness_cycles": ((f64_twin or {}).get("real_loop_cycles_per_block") if a.precision == "f64"
                                                               else use_bound.get("real_loop_cycles_per_block")) if use_bound else None,
                     "achieved_cycles_per_block_isolated_launch": achieved_cyc_iso,
                     "achieved_cycles_per_block_timed_region": achieved_cyc,
                     "frac": (bcyc / achieved_cyc_iso) if bcyc else None,
                     "frac_basis": "isolated launch (kernel_ms_avg: the quantity rocprofv3 --stats reports for --streams 1)",
                     "frac_timed_region": (bcyc / achieved_cyc) if bcyc else None,
                     "clock_mhz_nominal": info["clock_mhz"], "clock_mhz_in_kernel": clock_mhz,
                     "nominal_serial_model": {"cycles_per_block_2_4_8": cyc_per_block,
                                              "ratio_to_timed_region_at_nominal_clock": nominal_frac,
                                              "ratio_to_isolated_launch_at_nominal_clock": nominal_frac_iso,
                                              "note": "rounds 1-2 quoted these ratios as a roofline fraction; the model is NOT a bound (classes of "
                                                      "different waves partly co-issue: it was exceeded) and is kept for continuity only"}},
            "kernel_Mpaths_per_sec_isolated_launch": (2 * shard_n / kiso) / 1000.0,
            "serial": {"streams_per_gpu": 1, "ms_per_step": serial_ms, "value": (2 * shard_n / serial_ms) / 1000.0,
                       "vs_baseline": (2 * shard_n / serial_ms) / 1000.0 / V100_PUBLISHED_MPATHS,
                       "unit": "M paths/s per GPU", "on_a_caller_stream_ms_per_step": serial_caller_ms,
                       "on_a_caller_stream_value": (2 * shard_n / serial_caller_ms) / 1000.0,
                       "continuity": "rounds 1-2 reported `serial` on a caller's stream (= on_a_caller_stream_*); since round 3 `serial` is the "
                                     "context's own stream",
                       "note": "same K launches strictly one after another on the context's own stream, no all-reduce; median of %d regions.  "
                               "on_a_caller_stream: the same through a stream of the caller's, where every call ends with one hipEventRecord "
                               "(the stream-lifetime rule of include/hw_mc.h)" % R_side},
            "device": info["name"],
        }
        if "q3_vega_gpu" in extras and bcyc:
            # the Vega kernel runs the SAME time loop (ou_walk: identical instruction census, tests/test_kernel_census.py), one PATH
            # per thread where the ZBC kernel has one pair: the measured bound applies to it as it stands
            q3 = extras["q3_vega_gpu"]
            q3_cyc = q3.pop("kernel_ms_for_valu") * 1e-3 * mhz * 1e6 * info["compute_units"] * 4 / (((1 << 20) // 64) * (steps_exec / 4.0))
            q3["valu"] = {"achieved_cycles_per_block_isolated_launch": q3_cyc, "bound_cycles_per_block": bcyc, "frac": bcyc / q3_cyc}
        elif "q3_vega_gpu" in extras:
            extras["q3_vega_gpu"].pop("kernel_ms_for_valu", None)
        line.update(extras)
        if dist_on:
            line["multi_gpu"] = {"all_reduce": "5 f64 per pricing call" if B == 1 else "5 f64 x %d steps per message" % B,
                                 "world_size_seen": seen, "launch_skew": skew,
                                 "compute_only_ms_per_step": compute_only_ms,
                                 "exposed_allreduce_ms_per_step": ms_per_step - compute_only_ms,
                                 "exposed_note": "ms_per_step minus the same K steps in the same alternation WITHOUT the all-reduce (MAX over ranks, median of "
                                                 "%d barrier-bracketed regions): what the collectives add to a step after overlap" % R_side,
                                 "single_call_host_sum_ms": single_call_host_ms,
                                 "single_call_host_sum_note": "the same strictly serial pricing call with the five sums copied to the host and added by a gloo "
                                                              "all-reduce instead of RCCL (MAX over ranks)",
                                 "single_call_ms": single_call_ms,
                                 "single_call_value": (n_eff / single_call_ms) / 1000.0 if single_call_ms else None,
                                 "single_call_note": "one sharded pricing call end to end, strictly serial: launch -> all-reduce -> host synchronises (MAX over ranks)",
                                 "all_reduce_alone_ms": allreduce_alone_ms,
                                 "per_rank_kernel_ms": per_rank_kernel_ms,
                                 "end_to_end_ms_per_step": ms_per_step}
        if rehearsal:
            line["rehearsal"] = "gloo, all ranks on device 0 -- control-flow check only, not a measurement"
        chk.add("timed region: the reduced sums are finite and the mean price plausible",
                bool(0.0 < st["mean"] < 0.1 and all(np.isfinite(x).all() for x in sums)), "mean price %r" % st["mean"])
        if os.environ.get("HWMC_BENCH_TRACE"):
            print("bench: GPU legs done", file=sys.stderr, flush=True)
        if os.environ.get("HWMC_BENCH_REGIONS_FILE"):
            with open(os.environ["HWMC_BENCH_REGIONS_FILE"], "w") as fh:
                json.dump({"steps_per_region": K, "region_ms": [round(x * 1e3, 5) for x in elapsed_regions],
                           "region_ms_hip_events": [round(float(x), 5) for x in region_ms_events]}, fh)
        line["checks"] = {"passed": len(chk.passed), "failed": len(chk.failed), "failed_names": chk.failed, "passed_names": chk.passed,
                          "distance_in_standard_errors": chk.z,
                          "how": "closed forms = Jamshidian's bond-option price on the market curve and its sigma-derivative, the exact "
                                 "P(0,T); SE = the estimator's own standard error; 1e-13 = float64 summation order only"}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(P, f, a.profile, a.grid_steps, a.cpu_seconds)
        if dist_on:
            # LAST key for N > 1: the multi-GPU block, ending in the self-check against the committed one-GPU constants
            mg = line.pop("multi_gpu")
            mg["self_check"] = hwmc.parallel.self_check(dict(line, multi_gpu=mg), a.gpus)
            if world > 1:
                exit_code = 3 if mg["self_check"]["failed"] else exit_code
            line["multi_gpu"] = mg
        else:
            # LAST key for N = 1: every rate README.md quotes, with its ms and its fractions of the HBM and VALU-issue roofs
            claims = build_claims(line, chk.passed, chk.failed)
            line["claims"] = claims
        if chk.failed:
            exit_code = 3
        print(json.dumps(line), flush=True)
    for eng in engines:
        eng.close()
    if os.environ.get("HWMC_BENCH_TRACE"):
        print("bench: contexts closed", file=sys.stderr, flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        print("bench.py: the line above FAILED its own checks (checks.failed_names / multi_gpu.self_check.failed): exit %d" % exit_code, file=sys.stderr)
    return exit_code


if __name__ == "__main__":
    rc = main()
    if os.environ.get("HWMC_BENCH_TRACE"):
        print("bench: main returned", file=sys.stderr, flush=True)
    sys.exit(rc)
