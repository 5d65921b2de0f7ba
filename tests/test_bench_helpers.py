"""CPU tests of bench.py's pure helpers and defaults (no GPU: bench.main() itself refuses to run without one)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_union_of_dispatch_intervals(bench):
    assert bench.union_ms([]) == 0.0
    assert bench.union_ms([(0.0, 1.0), (2.0, 3.0)]) == pytest.approx(2.0)
    assert bench.union_ms([(0.0, 1.0), (0.5, 1.5), (1.5, 2.0), (5.0, 5.5)]) == pytest.approx(2.5)     # overlapping + touching + apart
    assert bench.union_ms([(2.0, 3.0), (0.0, 4.0)]) == pytest.approx(4.0)                              # nested, unsorted


def test_traffic_comes_from_the_newest_committed_pmc_summary(bench):
    traffic, src = bench.newest_pmc_summary(True)
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))
    assert files, "no rocprofv3 PMC summary committed"
    assert src["file"] == "profiles/" + files[-1]                          # rNN sorts by round
    with open(os.path.join(ROOT, src["file"])) as fh:
        d = json.load(fh)
    assert traffic == d["hbm_traffic_bytes_per_launch"]["corrected_bytes"]
    assert d["hbm_traffic_bytes_per_launch"]["corrected_bytes"] == pytest.approx(
        (2.0 * d["hbm_traffic_bytes_per_launch"]["FETCH_SIZE_KB"] + d["hbm_traffic_bytes_per_launch"]["WRITE_SIZE_KB"]) * 1024.0)
    assert bench.newest_pmc_summary(False) == (None, None)                 # any non-default configuration: no stale figure


def test_defaults_follow_the_driver_contract(bench, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps == 20 and a.warmup == 3                 # no flags: N = 1, finishes within minutes
    assert a.scaling == "strong" and a.total_log2 == 24 and a.allreduce_every == 1 and a.paths_log2 == 20
    # the VALU model bench.py prices the loop with is the one the census test pins
    assert bench.VALU_MODEL["superposed"] == {"full": 32, "half": 20, "quarter": 8}
    assert bench.VALU_MODEL["superposed_f64"] == {"full": 30, "half": 26, "quarter": 8}


def test_world_size_mismatch_and_missing_gpu_fail_loudly():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 2 and "no CPU fallback" in r.stderr and not r.stdout.strip()


def test_check_scaling_cli_on_driver_shaped_records(tmp_path):
    """tools/check_scaling.py takes the driver's records (the line under "parsed", several runs in one file) and runs
    hwmc.parallel.check_scaling_lines on every N > 1 line against the N = 1 line: here the committed N = 1 line and an 8-GPU
    line synthesised from it (same sums -> passes; one sum off by 1e-9 -> exit code 1 with the reason)."""
    import copy
    import glob
    d = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))[-1]))
    ref = d["strong_scaling_reference"]
    n8 = copy.deepcopy(d)
    n8.update(n_gpus=8, ms_per_step=0.36, value=2 * (1 << 24) / 0.36 / 1000.0, sums_step0=list(ref["sums_step0"]))
    n8["config"].update(pairs_per_step=1 << 24, effective_paths=2 << 24)
    n8["multi_gpu"] = {"world_size_seen": {"dist": 8, "allreduce_of_ones": 8}, "per_rank_kernel_ms": [0.355] * 8,
                       "exposed_allreduce_ms_per_step": 0.004, "launch_skew": {"median_ms": 0.003}, "single_call_ms": 0.41,
                       "single_call_host_sum_ms": 0.39}
    one, many, bad = (str(tmp_path / n) for n in ("bench.json", "scale.json", "bad.json"))
    json.dump({"parsed": d, "rc": 0}, open(one, "w"))
    json.dump({"runs": [{"n": 8, "parsed": n8}]}, open(many, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scaling.py"), one, many], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.splitlines()[-1])
    assert rep["n_gpus"] == 8 and rep["verdict"].startswith("scales") and rep["strong_scaling"] > 7.0
    n8["sums_step0"][2] *= 1.0 + 1e-9
    json.dump({"runs": [{"n": 8, "parsed": n8}]}, open(bad, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scaling.py"), one, bad], capture_output=True, text=True)
    assert r.returncode == 1 and "did not price the N = 1 sample set" in r.stdout


def test_tools_and_labs_compile():
    """every Python tool and lab at least byte-compiles (the labs are not run by any test: they are records of experiments)"""
    import glob
    import py_compile
    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "labs", "*.py"))
    assert len(files) > 30
    for f in files:
        py_compile.compile(f, doraise=True)


def test_roofline_report_reads_the_committed_line():
    """tools/roofline_report.py turns the committed bench line into the table DESIGN.md section 4 quotes: one row per measured
    launch with its HBM fraction and its achieved / bound cycles; every fraction it prints is within (0, 1.03]."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_report.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l for l in r.stdout.splitlines() if l.startswith(("ZBC", "pathwise", "FD Vega", "curve", "stress"))]
    assert len(rows) >= 12
    for l in rows:
        frac = float(l.split()[-2])
        assert 0.8 < frac <= 1.03, l
    assert "curve_f32_p1" in r.stdout and "zbc_f64" in r.stdout and "CPU baseline" in r.stdout
