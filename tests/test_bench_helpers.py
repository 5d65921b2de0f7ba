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
