"""GPU test of bench.py's command-line surface: the flag combinations the driver or a later round may use must run to
completion and print ONE JSON line carrying the contract's keys (no CPU baseline here: it is timed, not tested)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline"}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _bench(*args, env=None, raw=False):
    e = dict(os.environ)
    e.pop("HWMC_BENCH_PMC_FILE", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--spinup-ms", "50", *args],
                       capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return (json.loads(lines[0]), r.stdout) if raw else json.loads(lines[0])


@pytest.mark.parametrize("args,env", [
    (["--gpus", "1", "--steps", "4", "--warmup", "1"], None),
    (["--steps", "1", "--warmup", "0", "--regions", "1"], None),
    (["--steps", "3", "--streams", "1", "--no-extras"], None),
    (["--steps", "3", "--streams", "3", "--paths-log2", "18", "--grid-steps", "4000", "--no-extras"], None),
    (["--steps", "3", "--precision", "f64", "--no-extras"], None),
    (["--steps", "3", "--stepper", "reference", "--profile", "readme", "--no-extras"], None),
    (["--steps", "5", "--warmup", "0", "--allreduce-every", "2", "--no-extras"], {"HWMC_BENCH_FORCE_RCCL": "1"}),
    (["--steps", "4", "--no-extras"], {"HWMC_BENCH_FORCE_RCCL": "1"}),
    (["--steps", "4", "--allreduce-every", "0", "--scaling", "weak", "--no-extras"], {"HWMC_BENCH_FORCE_RCCL": "1"}),
])
def test_bench_flag_combinations(args, env):
    d = _bench("--regions", "3", *args, env=env) if "--regions" not in args else _bench(*args, env=env)
    assert CONTRACT_KEYS <= set(d), sorted(CONTRACT_KEYS - set(d))
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] in ("strong", "weak") and d["data"] == "synthetic" and d["unit"] == "M paths/s"
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_avg",
                                  "kernel_ms_per_step_in_timed_region", "launch_geometry"}
    assert 0.0 < d["price_mean"] < 0.1 and d["config"]["workload"].startswith("Q2b option_pricing")
    assert d["steps"] == int(args[args.index("--steps") + 1])
    sp = d["value_spread"]
    assert sp["min"] <= sp["p10"] <= sp["median"] <= sp["p90"] <= sp["max"] and sp["median"] == d["value"] and "region_ms" not in sp
    assert sp["region_ms_min_median_max"][1] == pytest.approx(d["ms_per_step"] * d["steps"], rel=1e-3)
    # the line checks itself: nothing failed (a failure would also have made bench.py exit 3), traffic is a measurement or null
    assert d["checks"]["failed"] == 0 and d["checks"]["passed"] >= 2 and d["checks"]["failed_names"] == []
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_source"]["measured_in_this_run"] is False
    assert d["config"]["counting"] == ("%d queues in flight" % d["config"]["streams_per_gpu"] if d["config"]["streams_per_gpu"] > 1 else "strictly serial")
    if env:
        assert list(d)[-1] == "multi_gpu" and list(d["multi_gpu"])[-1] == "self_check"
        sc = d["multi_gpu"]["self_check"]
        assert sc["failed"] == [] and sc["world_size_seen"] == {"dist": 1, "allreduce_of_ones": 1}
        if "weak" not in args:                             # 2^20 pairs, seed 1234: the committed constant of BASELINE configs[1]
            assert sc["reference"] == "zbc|head|f32|superposed|1000|20|1234" and sc["sums_max_rel_diff"] <= 1e-13
    else:
        assert list(d)[-1] == "claims" and d["claims"]["checks_failed"] == [] and d["claims"]["checks_passed"] == d["checks"]["passed"]
        assert d["vs_baseline_like_for_like"] == pytest.approx(d["kernel_Mpaths_per_sec_isolated_launch"] / 1064.0, rel=1e-3)
    # the kernels cannot occupy the chip for longer per step than a step lasts (2 % for timer granularity)
    assert d["roofline"]["kernel_ms_per_step_in_timed_region"] <= 1.02 * d["ms_per_step"]
    g = d["roofline"]["launch_geometry"]
    assert g["grid"] * g["block"] * g["iters"] * g["units_per_thread"] >= d["config"]["pairs_per_gpu"] and g["units_per_thread"] in (1, 2)
    assert d["roofline"]["algorithmic_bytes_per_launch"] == g["partial_bytes"] + 40
    if env:
        assert d["multi_gpu"]["single_call_ms"] > 0 and len(d["multi_gpu"]["per_rank_kernel_ms"]) == 1


def test_bench_default_line_carries_the_side_measurements():
    """No flags beyond a short K: the N = 1 line must hold the other stepper, the 20-run single launch, the Q1 curve kernel
    at configs[0]'s size, the 2^24-pair strong-scaling reference, the 10 K-step region and the same K calls through the
    library's own pipe (whose sums bench.py itself compares with the timed region's), all as throughputs in the metric's unit."""
    d, stdout = _bench("--steps", "6", "--regions", "3", raw=True)
    # the line ends with `claims`, <= 1.5 KB, so the last 2 KB of the output -- all that a driver record keeps -- hold every
    # README row, the size table, the stress launches and the outcome of the checks
    tail = stdout.rstrip()[-2048:]
    assert '"claims": {' in tail and len(json.dumps(d["claims"])) <= 1536
    cl = d["claims"]
    assert list(d)[-1] == "claims" and cl["checks_failed"] == [] and cl["checks_passed"] >= 20, d["checks"]
    for k in ("kernel_alone", "serial", "two_queues", "sync_call", "library_pipeline", "long_region", "runs20_single_launch", "other_stepper",
              "q1_curve_gpu", "q3_vega_gpu", "q3_fd_vega_gpu"):
        assert cl["rates"][k][0] > 1064.0 * (1 if k.startswith("q") else 5) and cl["rates"][k][1] > 0, k
    assert cl["rates"]["sync_call"][0] == pytest.approx(d["sync_call"]["value"], rel=1e-5) and cl["rates"]["q1_curve_gpu"][2] == pytest.approx(d["q1_curve_gpu"]["valu"]["frac"], abs=1e-4)
    assert [r[0] for r in cl["size_table"]] == [20, 22, 24, 26] and all(0 < r[2] < 1e-4 and 0.5 < r[3] <= 1.03 for r in cl["size_table"])
    for prec in ("f32", "f64"):
        for k in ("zbc", "curve"):
            assert cl["stress"][prec][k][0] == pytest.approx(d["stress"][prec][k]["kernel_ms"], rel=1e-4) and 0 < cl["stress"][prec][k][2] < 1e-4
    assert cl["vs_baseline_counting"] == "2 queues in flight" and cl["vs_baseline_like_for_like"] == d["vs_baseline_like_for_like"] < cl["vs_baseline"] == d["vs_baseline"]
    # every timed side launch looked at its own output: the names say what was compared
    names = " | ".join(d["checks"]["passed_names"])
    for must in ("stress.f32.zbc: price within 4 SE", "stress.f64.curve: path count exact", "stress.curve: float and double agree",
                 "q1_curve_gpu: path count exact", "q3_vega_gpu: pathwise Vega within 4 SE", "q3_fd_vega_gpu: finite-difference Vega",
                 "size_table 2^26: price within 4 SE", "sync_call:", "runs20_single_launch: per-run sums bit-identical",
                 "library_pipeline: hw_pipe_zbc_moments returns", "other_stepper: the two steppers agree", "strong_scaling_reference: equals the committed"):
        assert must in names, (must, d["checks"])
    for key in ("other_stepper", "runs20_single_launch", "q1_curve_gpu", "q3_vega_gpu", "strong_scaling_reference", "serial", "long_region", "library_pipeline"):
        assert d[key]["value"] > 0, key
    # the three rates the reference publishes (README.md:161-163), each in its own metric: Q2b = value, Q1, Q3
    assert d["q3_vega_gpu"]["paths"] == 1 << 20 and d["q3_vega_gpu"]["value"] > 509.0 and d["q3_vega_gpu"]["vs_v100_509"] > 1.0
    # the VALU reference rate is MEASURED (tools/mix_bound.hip, live or the committed profile): the dependency-free rate of the
    # production ordering -- the fastest ordering tried, not a proven floor (that one is exclusive_classes_floor) -- and both it and
    # the launch time are live measurements, so the comparison carries a 3 % tolerance instead of a strict inequality
    v = d["valu"]
    assert v["bound_cycles_per_block"] and 185.0 < v["bound_cycles_per_block"] < 200.0, v
    assert 0.5 < v["frac"] <= 1.03 and 0.5 < v["frac_timed_region"] <= 1.03, v
    assert v["exclusive_classes_floor"]["nominal_4_4_8"] == 144 and v["achieved_cycles_per_block_isolated_launch"] > 0.97 * v["bound_cycles_per_block"]
    assert v["achieved_cycles_per_block_isolated_launch"] > v["exclusive_classes_floor"]["nominal_4_4_8"]        # the provable floor
    # BASELINE configs[2]'s finite-difference half, the 2^20..2^26 size table, the synchronous-call rate and configs[4] on one GPU
    fd = d["q3_fd_vega_gpu"]
    assert fd["scenarios"] == 3 and fd["value"] > 1064.0 and 0.15 < fd["vega_fd"] < 0.35 and 0.5 < fd["valu"]["frac"] <= 1.03
    rows = d["size_table"]["rows"]
    assert [r["pairs_log2"] for r in rows] == [20, 22, 24, 26] and all(r["value"] > 1064.0 and 0.5 < r["valu_frac"] <= 1.03 for r in rows)
    assert rows[-1]["value"] > rows[0]["value"]                          # larger batches amortise the launch's fixed costs
    sc = d["sync_call"]
    assert sc["calls"] == 20 and sc["ms_per_call"] >= d["roofline"]["kernel_ms_min"] and sc["value"] > 1064.0
    st = d["stress"]
    for prec in ("f32", "f64"):
        for k, loop in (("zbc", "zbc_" + prec), ("curve", "curve_%s_p2" % prec)):
            x = st[prec][k]
            assert x["value"] > 0 and x["hbm"]["frac"] < 1e-3 and x["valu"]["loop"] == loop and 0.5 < x["valu"]["frac"] <= 1.03, (prec, k, x)
    assert 1.0 < st["f64_over_f32"]["zbc"] < 1.3 and 1.0 < st["f64_over_f32"]["curve"] < 1.3
    assert d["q1_curve_gpu"]["valu"]["loop"] == "curve_f32_p1" and 0.5 < d["q1_curve_gpu"]["valu"]["frac"] <= 1.03
    assert len(d["strong_scaling_reference"]["sums_step0"]) == 5 and d["strong_scaling_reference"]["kernel_ms"] > 0
    assert d["timed_region_s"] > 0
    assert d["long_region"]["steps"] == 60 and d["library_pipeline"]["api"] == "hw_pipe_zbc_moments" and d["library_pipeline"]["steps"] == 6
    assert d["strong_scaling_reference"]["pairs_per_step"] == 1 << 24 and d["config"]["pairs_per_step"] == 1 << 20
    assert d["scaling"] == "strong" and d["config"]["effective_paths"] == 2 << 20
    assert d["runs20_single_launch"]["value"] > 0.9 * d["serial"]["value"]
    assert d["q1_curve_gpu"]["value"] > 391.0                  # the reference's V100 figure for this kernel (README.md:9-12)


@pytest.mark.parametrize("every,regions", [("1", "2"), ("0", "2"), ("1", None)])
def test_bench_two_rank_strong_scaling_rehearsal(every, regions):
    """N = 2 launched exactly as the driver does (torch.distributed.run, one process per rank), in rehearsal mode: both
    ranks on the one visible GPU, all-reduce over gloo.  Not a measurement -- it pins the multi-rank control flow of
    the DEFAULT mode: strong scaling of one 2^k-pair pricing call (BASELINE configs[3] at k = 24; k = 18 here), sharded
    by global pair id, ONE all-reduce per pricing call, MAX of the elapsed time over ranks, rank 0 prints ONE line --
    and that the reduced sums are the sums one process computes for the whole call, to 1e-13."""
    import numpy as np
    import hwmc
    e = dict(os.environ, HWMC_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup-ms", "50",
                        *(["--regions", regions] if regions else []),   # None: the driver's form -- the region count is agreed between the ranks
                        "--total-log2", "18", "--allreduce-every", every],
                       capture_output=True, text=True, timeout=400, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(d) and d["n_gpus"] == 2 and "rehearsal" in d
    if regions is None:
        assert 7 <= d["value_spread"]["timed_regions"] <= 400 and d["timed_region_s"] > 0
        assert len(d["value_spread"]["region_ms_min_median_max"]) == 3 and "region_ms" not in d["value_spread"]
    assert d["scaling"] == "strong"
    assert d["config"]["pairs_per_step"] == 1 << 18 and d["config"]["pairs_per_gpu"] == 1 << 17
    assert d["config"]["effective_paths"] == 2 * (1 << 18)
    assert d["config"]["allreduce_every"] == int(every)
    assert "one collective per pricing call" in d["config"]["parallelism"] if every == "1" else "3 steps per message" in d["config"]["parallelism"]
    p = hwmc.default_params("head")
    P, f = hwmc.analytic_curve(p)
    with hwmc.HullWhiteEngine(p, device=0) as eng:
        eng.load_market_data(P, f)
        whole = eng.simulate_ZBC_control_variate(1 << 18, 1234)        # the timed steps' first seed (2_option:229)
    np.testing.assert_allclose(d["sums_step0"], whole, rtol=1e-13)
    assert abs(d["price_mean"] - 0.0252555) < 5e-4
    mg = d["multi_gpu"]
    assert len(mg["per_rank_kernel_ms"]) == 2 and mg["single_call_ms"] > 0
    # the line ends with the multi-GPU block and that with its self-check: both ranks seen, no committed constant for a 2^18-pair
    # call (reported, not failed: the run exits 0), and the check against the committed 2^24-pair constants is exercised below
    assert list(d)[-1] == "multi_gpu" and list(mg)[-1] == "self_check"
    sc = mg["self_check"]
    assert sc["failed"] == [] and sc["gpus_requested"] == 2 and sc["world_size_seen"] == {"dist": 2, "allreduce_of_ones": 2}
    assert sc["sums_max_rel_diff"] is None and "no committed one-GPU constant" in sc["verdict"] and sc["reference"] == "zbc|head|f32|superposed|1000|18|1234"
    # the line checks itself: both ranks took part, the ranks' barrier release is timed, the same K steps
    # were also run without the collective, and the host-sum form of one call sits beside the collective's
    assert mg["world_size_seen"] == {"dist": 2, "allreduce_of_ones": 2}
    assert mg["launch_skew"]["regions"] >= 1 and 0.0 <= mg["launch_skew"]["median_ms"] < 100.0
    assert mg["compute_only_ms_per_step"] > 0 and mg["single_call_host_sum_ms"] > 0
    assert mg["exposed_allreduce_ms_per_step"] == pytest.approx(d["ms_per_step"] - mg["compute_only_ms_per_step"], abs=2e-5)   # (ms_per_step is rounded to 5 places in the line)
    if every == "1" and regions is None:
        # ... and against the N = 1 line of the same build: the sample set must be the same to 1e-13
        one = _bench("--steps", "3", "--regions", "2", "--total-log2", "18", env={"HWMC_BENCH_NO_STRESS": "1"})
        rep = hwmc.parallel.check_scaling_lines(one, d)
        assert rep["sums_step0_max_rel_diff"] <= 1e-13 and rep["n_gpus"] == 2 and "verdict" in rep
        tampered = json.loads(json.dumps(d))
        tampered["sums_step0"][0] *= 1.0 + 1e-9
        with pytest.raises(AssertionError, match="did not price the N = 1 sample set"):
            hwmc.parallel.check_scaling_lines(one, tampered)


def test_bench_two_rank_line_checks_itself_against_the_committed_constants(tmp_path):
    """VERDICT r4 item 3, on the one GPU there is: two ranks (rehearsal: both on device 0, gloo) share ONE 2^20-pair pricing call,
    whose whole-call sums are a committed constant (reference_sums.json, BASELINE configs[1]'s call) -- multi_gpu.self_check
    must find the all-reduced sums of the first timed step equal to it to 1e-13 and say so; against a store whose constant is
    off by 1e-9 the same run prints its line with the reason and exits with code 3."""
    import hwmc

    def run(env_extra):
        e = dict(os.environ, HWMC_BENCH_REHEARSAL="1", **env_extra)
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                               "--gpus", "2", "--steps", "2", "--warmup", "1", "--spinup-ms", "50", "--regions", "2", "--total-log2", "20"],
                              capture_output=True, text=True, timeout=400, env=e, cwd=ROOT)
    r = run({})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    sc = d["multi_gpu"]["self_check"]
    assert sc["reference"] == "zbc|head|f32|superposed|1000|20|1234" and sc["sums_max_rel_diff"] <= 1e-13 and sc["failed"] == []
    assert d["config"]["pairs_per_gpu"] == 1 << 19 and sc["world_size_seen"] == {"dist": 2, "allreduce_of_ones": 2}
    store = dict(hwmc.parallel.load_reference_sums())
    store[sc["reference"]] = [x * (1.0 + (1e-9 if i == 2 else 0.0)) for i, x in enumerate(store[sc["reference"]])]
    bad = tmp_path / "reference_sums.json"
    bad.write_text(json.dumps(store))
    r = run({"HWMC_REFERENCE_SUMS": str(bad)})
    assert r.returncode != 0 and "FAILED its own checks" in r.stderr, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])          # the line is still printed, with the reason
    sc = d["multi_gpu"]["self_check"]
    assert len(sc["failed"]) == 1 and "another sample set" in sc["failed"][0] and sc["verdict"].startswith("FAILED")
    assert 5e-10 < sc["sums_max_rel_diff"] < 2e-9
