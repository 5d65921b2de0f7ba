"""CPU test of the C++ host logic of the executables (src/hw_app.hpp) and of the host-only C-ABI entry
points, without a GPU: sharding, 20-run statistics, control-variate post-processing, theta recovery and the
reference's file formats (JSON header/array/performance blocks, CSV, raw float32 arrays)."""
import json
import re
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "monte-carlo-simulation-of-hull-white-model-and-sensitivities-computation_amd")


@pytest.fixture(scope="module")
def probe(tmp_path_factory, hwmc_mod):
    out = tmp_path_factory.mktemp("probe")
    exe = str(out / "host_logic_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_logic_probe.cpp"), "-o", exe,
                           "-L" + PKG, "-lhw_hip", "-Wl,-rpath," + PKG])
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][0]
    return json.loads(line), out


def test_cpp_sharding_statistics_and_cv(probe, hwmc_mod):
    d, _ = probe
    assert d["shards"] == [[0, 334], [334, 334], [668, 333]]
    assert [list(hwmc_mod.parallel.shard(1001, g, 3)) for g in range(3)] == d["shards"]     # same partition as Python
    x = np.array([0.0354 + 1e-5 * ((i * 7) % 11) for i in range(20)])
    st = hwmc_mod.run_statistics(x)
    assert d["stats"]["mean"] == pytest.approx(st["mean"], rel=1e-14)
    assert d["stats"]["sd"] == pytest.approx(x.std(ddof=1), rel=1e-12)
    assert d["stats"]["margin"] == pytest.approx(2.093 * x.std(ddof=1) / np.sqrt(20), rel=1e-12)   # 2_option:320
    assert d["cv"]["beta"] == pytest.approx(1.0) and d["cv"]["price"] == pytest.approx(0.9)
    assert d["theta_max_err"] < 0.01 and d["P100"] == pytest.approx(0.876777, abs=2e-6) and d["f0"] == pytest.approx(0.012, abs=1e-7)


def test_cpp_writers_follow_the_reference_formats(probe, hwmc_mod):
    _, out = probe
    js = json.load(open(out / "probe.json"))
    assert list(js.keys()) == ["task", "timestamp", "parameters", "P", "performance"]          # output.cuh:19-31
    assert js["parameters"] == {"N_PATHS": 1 << 20, "N_STEPS": 1000, "N_MAT": 101, "T_FINAL": 10.0, "a": 1.0,
                                "sigma": 0.1, "r0": 0.012}
    assert len(js["P"]) == 101 and js["P"][0] == 1.0
    assert js["performance"] == {"simulation_time_ms": 0.25, "throughput_Mpaths_per_sec": 8388.61}   # (n/ms)/1000, %.2f
    text = open(out / "probe.json").read()
    assert '  "P": [\n    1.00000000, ' in text                                                # ten %.8f values per line
    csv = open(out / "probe.csv").read().splitlines()
    assert csv[0] == "T,P(0 T)" and csv[1] == "0.0000,1.00000000" and len(csv) == 102          # output.cuh:70-73
    raw = np.fromfile(out / "probe.bin", dtype=np.float32)
    P, _ = hwmc_mod.analytic_curve(hwmc_mod.default_params("readme"))
    assert np.array_equal(raw, P)                                                              # common.cuh:144-153


def test_integration_md_call_shapes_compile_and_link(tmp_path, hwmc_mod):
    """Every call sequence INTEGRATION.md shows a maintainer of the reference (Q1, Q2b, the 20-run harnesses, Q3,
    the in-process communicator) compiles against include/hw_mc.h and links against libhw_hip.so (nothing is run:
    there is no GPU here)."""
    exe = str(tmp_path / "integration_shapes")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-Wno-unused-variable", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "integration_shapes.cpp"), "-o", exe,
                           "-L" + PKG, "-lhw_hip", "-Wl,-rpath," + PKG])
    assert os.path.exists(exe)


def test_stream_guard_state_machine_against_a_mock_runtime(tmp_path):
    """The guard that orders a context's launches across streams (csrc/hw_stream_guard.hpp) run against a mock of HIP's
    stream/event semantics with caller streams created and destroyed at random (tests/cpp/stream_guard_probe.cpp):
    every enqueue ordered behind all earlier ones, no record/wait ever on a dead stream, a pipe's contexts record only at
    switches -- and round 2's lazy scheme, on the same schedule, does touch dead streams (the use-after-free behind
    GPUTEST_r02.json's SIGSEGV).  A third pass makes one record in fifty FAIL: the guard drains that stream on the host,
    forgets it and goes on (no later call refused, nothing left pending on a stream it cannot name)."""
    exe = str(tmp_path / "stream_guard_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(PKG, "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "stream_guard_probe.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["order_violations"] == 0 and d["calls"] > 150_000 and d["calls_on_caller_streams"] > 30_000 and d["switches"] > 100_000
    assert d["lazy_scheme_dead_stream_uses"] > 0
    assert "0 uses of a dead stream" in r.stdout and "LIFETIME violated" not in r.stdout
    assert re.search(r"mode 2: \d+ records \((\d+) failed -> \1 host syncs\)", r.stdout), r.stdout


def test_fleet_two_phase_protocol_against_a_mock_collective(tmp_path):
    """hw_comm_*'s host machinery (csrc/hw_fleet.hpp: persistent workers, rendezvous between a device's own work and the
    collective, abort for failures inside it) with 1, 2, 3 and 8 workers against a mock all-reduce that -- like the real
    one -- does not return until every rank has entered it (tests/cpp/fleet_probe.cpp): no rank enters the collective unless
    all ranks got through phase 1, a phase-1 failure names its rank and leaves the fleet usable, a phase-2 failure aborts
    exactly once and releases everybody, nothing ever waits longer than the watchdog.  The multi-GPU path has never run on
    more than one device (no multi-GPU box): this is the part of it that does not need one."""
    exe = str(tmp_path / "fleet_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", os.path.join(PKG, "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "fleet_probe.cpp"), "-o", exe])
    r = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["wrong"] == 0 and d["slow_dispatches"] == 0 and d["checks"] > 15_000 and d["phase1_failures"] > 1_500
    assert d["workers_bound"] == 1 + 2 + 3 + 8
