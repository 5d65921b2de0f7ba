"""GPU test of the process surface: bin/q1, bin/q2, bin/q3 run from a scratch directory with the
reference's ordering dependency (q2/q3 need q1's data/P.bin, data/f.bin) and write the file set and
schemas the reference's analyze.py consumes (analyze.py:15-27, 64-83, 105-123, 146-153, 257-342)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(exe, cwd, *args, stdin=None, check=True):
    r = subprocess.run([os.path.join(ROOT, "bin", exe), *args], cwd=cwd, input=stdin, capture_output=True, text=True, timeout=600)
    if check:
        assert r.returncode == 0, r.stdout + r.stderr
    return r


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", ROOT, "all", "benchmark", "throughput"], stdout=subprocess.DEVNULL)


def test_q2_without_q1_exits_like_the_reference(built, tmp_path):
    r = _run("q2", tmp_path, "--no", check=False)
    assert r.returncode == 1
    assert "Did you run Q1 first?" in r.stdout                   # common.cuh:155-161


def test_three_executables_file_contract(built, tmp_path):
    d = tmp_path
    r1 = _run("q1", d, "--paths-log2", "18", "--seed", "1234", "--theta-profile", "readme")
    for line in ("P(0,0) = 1.0:", "P(0,10) ~ 0.87:", "Throughput:", "Saved data/P.bin (101 floats)"):
        assert line in r1.stdout
    assert "ERROR" not in r1.stdout
    P = np.fromfile(d / "data" / "P.bin", dtype=np.float32)
    f = np.fromfile(d / "data" / "f.bin", dtype=np.float32)
    assert P.size == 101 and f.size == 101 and P[0] == 1.0
    assert abs(P[100] - 0.876777) < 2e-3 and abs(f[0] - 0.012) < 2e-3
    paths = np.fromfile(d / "data" / "r_paths.bin", dtype=np.float32)
    assert paths.size == 32 * 1001 and np.all(paths.reshape(32, 1001)[:, 0] == np.float32(0.012))
    q1 = json.load(open(d / "data" / "q1_results.json"))
    assert q1["parameters"]["N_PATHS"] == 1 << 18 and q1["parameters"]["N_STEPS"] == 1000
    assert set(q1["validation"]) == {"P_0_0", "P_0_10", "f_0_0"} and len(q1["P"]) == 101 and len(q1["f"]) == 101
    assert q1["performance"]["simulation_time_ms"] > 0 and q1["performance"]["throughput_Mpaths_per_sec"] > 0
    assert open(d / "data" / "P_curve.csv").readline().strip() == "T,P(0 T)"
    assert open(d / "data" / "f_curve.csv").readline().strip() == "T,f(0 T)"
    assert "HULL-WHITE MODEL SIMULATION RESULTS" in open(d / "data" / "summary.txt").read()

    # q2: answers "y" on stdin like an interactive user (2_option:495-500)
    r2 = _run("q2", d, "--paths-log2", "18", "--seed", "99", "--theta-profile", "readme", stdin="y\n")
    assert "Recovery: SUCCESS" in r2.stdout and "Beta optimal:" in r2.stdout
    q2a = json.load(open(d / "data" / "q2a_results.json"))
    assert q2a["error_metrics"]["success"] is True and q2a["error_metrics"]["max_error"] < 0.01
    q2b = json.load(open(d / "data" / "q2b_results.json"))
    assert abs(q2b["results"]["ZBC_control_variate"] - 0.0354) < 1.5e-3
    assert "control_deviation" in q2b["results"] and q2b["performance"]["throughput_Mpaths_per_sec"] > 0
    assert open(d / "data" / "theta_comparison.csv").readline().strip() == "T,theta_original,theta_recovered"
    boot = open(d / "data" / "zbc_bootstrap_optimal.csv").read().splitlines()
    assert boot[0] == "run,price_adjusted,price_raw,beta_optimal,correlation" and len(boot) == 21
    assert "Variance Reduction:" in open(d / "data" / "zbc_statistics_optimal.txt").read()
    assert "Theta recovery: SUCCESS (max error < 0.01)" in open(d / "data" / "summary.txt").read()

    # q3: "n" to the sweep, "y" to the validation (3_sens:725-749)
    r3 = _run("q3", d, "--paths-log2", "18", "--seed", "5", "--theta-profile", "readme", stdin="n\ny\n")
    assert "Sign Check:                 PASS" in r3.stdout and "Magnitude Check:            PASS" in r3.stdout
    q3 = json.load(open(d / "data" / "q3_results.json"))
    res = q3["results"]
    assert set(res) == {"sensitivity_mc", "sensitivity_fd", "abs_diff"}
    assert abs(res["sensitivity_mc"] - 0.2302) < 6e-3 and abs(res["sensitivity_fd"] - 0.2302) < 8e-3
    vb = open(d / "data" / "vega_bootstrap.csv").read().splitlines()
    assert vb[0] == "run,vega" and len(vb) == 21
    assert os.path.exists(d / "data" / "vega_statistics.txt")
    summary = open(d / "data" / "summary.txt").read()
    assert "Q1: ZERO-COUPON BOND PRICING" in summary and "Q2: THETA RECOVERY & OPTION PRICING" in summary and "Q3: SENSITIVITY ANALYSIS" in summary


def test_q3_block_size_sweep_path(built, tmp_path):
    """q3 answering "y" to the first prompt runs benchmark_block_sizes (src/3_sensitivity_analysis.cu:277-371): one
    warm-up + 20 timed launches of the pathwise-Vega kernel per workgroup size.  The reference's four sizes must each
    give a row (plus 64, one wavefront), every row with a positive time, and the run must go on to the FD section and
    leave the launch geometry on automatic (the Vega printed afterwards by the validation equals a default run's)."""
    _run("q1", tmp_path, "--paths-log2", "16", "--seed", "21", "--theta-profile", "readme", "--spinup-ms", "0")
    r = _run("q3", tmp_path, "--paths-log2", "16", "--seed", "5", "--theta-profile", "readme", "--spinup-ms", "0", stdin="y\nn\n")
    out = r.stdout
    assert "---BLOCK SIZE OPTIMIZATION SWEEP---" in out and "Best configuration:" in out and "Relative Performance:" in out
    rows = {}
    for line in out.splitlines():
        parts = [x.strip() for x in line.split("|")]
        if len(parts) == 5 and parts[0].isdigit():
            rows[int(parts[0])] = (int(parts[1]), float(parts[2]), float(parts[3]))
    assert set(rows) == {64, 128, 256, 512, 1024}                       # reference: {128, 256, 512, 1024} (3_sens:286)
    for bs, (grid, ms, thr) in rows.items():
        assert grid == ((1 << 16) + bs - 1) // bs and ms > 0 and thr > 0
    assert out.index("Best configuration:") < out.index("FINITE DIFFERENCE APPROXIMATION")
    q3 = json.load(open(tmp_path / "data" / "q3_results.json"))
    r0 = _run("q3", tmp_path, "--paths-log2", "16", "--seed", "5", "--theta-profile", "readme", "--spinup-ms", "0", "--no")
    q3n = json.load(open(tmp_path / "data" / "q3_results.json"))
    assert q3["results"] == q3n["results"]                               # the sweep changed no result
    assert "OPTIMAL" in out and "ERROR" not in r0.stdout


def test_exact_market_and_reference_stepper_flags(built, tmp_path):
    r = _run("q2", tmp_path, "--paths-log2", "16", "--seed", "3", "--market", "exact", "--stepper", "reference",
             "--precision", "f32", "--no")
    assert "Market curve: exact" in r.stdout and "ZBC (control variate adjusted):" in r.stdout


def test_rccl_reduce_flag_gives_the_same_files(built, tmp_path):
    """--reduce rccl routes the cross-device sums through hw_comm_* (a one-rank RCCL communicator on this
    box); curve and price must equal the default host-side path digit for digit."""
    a, b = tmp_path / "host", tmp_path / "rccl"
    a.mkdir(); b.mkdir()
    common = ("--paths-log2", "16", "--seed", "11", "--theta-profile", "readme")
    _run("q1", a, *common)
    r = _run("q1", b, *common, "--reduce", "rccl")
    assert "Cross-device sums: RCCL all-reduce over 1 device" in r.stdout
    assert np.array_equal(np.fromfile(a / "data" / "P.bin", np.float32), np.fromfile(b / "data" / "P.bin", np.float32))
    _run("q2", a, *common, "--no")
    _run("q2", b, *common, "--reduce", "rccl", "--no")
    ja, jb = (json.load(open(d / "data" / "q2b_results.json")) for d in (a, b))
    assert ja["results"] == jb["results"]
    _run("q3", a, *common, "--no")
    r3 = _run("q3", b, *common, "--reduce", "rccl", "--no")
    assert "ERROR" not in r3.stdout


def test_reduction_benchmark_target(built, tmp_path):
    """bin/benchmark (the reference's optional `make benchmark`, src/benchmark_reductions.cu): needs q1's curves, runs
    the production path loop with five ways of finishing the sums, all of which must give the same price, and writes
    data/benchmark_reductions.json in the schema the reference's analyze.py reads (analyze.py:195-203)."""
    r = _run("benchmark", tmp_path, check=False)
    assert r.returncode == 1 and "Did you run Q1 first?" in r.stdout
    _run("q1", tmp_path, "--paths-log2", "16", "--seed", "5", "--theta-profile", "readme", "--spinup-ms", "0")
    r = _run("benchmark", tmp_path, "--paths-log2", "18", "--theta-profile", "readme")
    assert "All methods produce identical results" in r.stdout
    j = json.load(open(tmp_path / "data" / "benchmark_reductions.json"))
    assert j["parameters"]["N_PATHS"] == 1 << 18 and len(j["results"]) == 5
    prices = {x["price"] for x in j["results"]}
    assert len(prices) == 1 and abs(prices.pop() - 0.0354) < 2e-3
    for x in j["results"]:
        assert set(x) == {"method", "time_ms", "throughput_Mpaths_per_sec", "price"} and x["time_ms"] > 0


def test_throughput_driver_over_the_c_abi(built, tmp_path):
    """bin/throughput: the headline metric from plain C++ (no Python, no torch in the process): 20 pricing calls of 2^20
    pairs made synchronously and through hw_pipe_* with two queues in flight; the program itself compares the sums bit
    for bit.  Only the ORDERING of the two rates is asserted here; the absolute bar (10 640 M paths/s = 10x the V100) is a
    perf claim and lives in tests/test_perf_claims.py (-m perf), so a slow box cannot fail the correctness suite."""
    r = _run("throughput", tmp_path, "--theta-profile", "readme")
    assert "sums identical: yes" in r.stdout and "FAILED" not in r.stdout
    j = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][-1])
    assert j["calls"] == 20 and j["queues"] == 2 and j["pairs_per_call"] == 1 << 20
    assert j["pipelined"] > j["synchronous"] > 0.0, j                        # measured 12 100-12 170 and 10 300-10 430
    price = float(r.stdout.split("ZBC price of call 0:")[1].split()[0])
    assert abs(price - 0.03544557) < 2e-4                                   # Jamshidian, readme profile (one run of 2^20 pairs)
    small = _run("throughput", tmp_path, "--paths-log2", "12", "--calls", "3", "--queues", "3", "--batches", "2", "--precision", "f64")
    assert "sums identical: yes" in small.stdout
    assert _run("throughput", tmp_path, "--calls", "0", check=False).returncode == 2
