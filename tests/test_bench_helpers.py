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


def test_traffic_is_reported_only_when_measured_on_this_box(bench, monkeypatch, tmp_path):
    """roofline.traffic is a PMC measurement or null -- never a number copied from another machine.  Without
    HWMC_BENCH_PMC_FILE (set by tools/profile_round.sh after its own --pmc passes on the box) traffic is None and
    traffic_source only NAMES the newest committed summary; with it, the file's corrected byte count is reported."""
    monkeypatch.delenv("HWMC_BENCH_PMC_FILE", raising=False)
    traffic, src = bench.pmc_traffic(True)
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))
    assert files, "no rocprofv3 PMC summary committed"
    assert traffic is None and src["measured_in_this_run"] is False
    assert src["committed_reference"] == "profiles/" + files[-1]           # rNN sorts by round
    with open(os.path.join(ROOT, src["committed_reference"])) as fh:
        d = json.load(fh)
    assert src["committed_reference_bytes_per_launch"] == d["hbm_traffic_bytes_per_launch"]["corrected_bytes"]
    assert d["hbm_traffic_bytes_per_launch"]["corrected_bytes"] == pytest.approx(
        (2.0 * d["hbm_traffic_bytes_per_launch"]["FETCH_SIZE_KB"] + d["hbm_traffic_bytes_per_launch"]["WRITE_SIZE_KB"]) * 1024.0)
    here = tmp_path / "pmc_here.json"
    here.write_text(json.dumps({"kernel": "zbc_superposed_kernel", "command": "bench.py", "hbm_traffic_bytes_per_launch": {"corrected_bytes": 123456.0}}))
    monkeypatch.setenv("HWMC_BENCH_PMC_FILE", str(here))
    traffic, src = bench.pmc_traffic(True)
    assert traffic == 123456.0 and src["measured_in_this_run"] is True and src["file"] == str(here)
    assert bench.pmc_traffic(False)[0] is None                             # any non-default configuration: no figure at all
    monkeypatch.setenv("HWMC_BENCH_PMC_FILE", str(tmp_path / "missing.json"))
    traffic, src = bench.pmc_traffic(True)
    assert traffic is None and "unreadable" in src["error"]


def _committed_line():
    import glob
    import re
    best = max(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")), key=lambda f: int(re.search(r"r(\d+)_bench", f).group(1)))
    with open(best) as fh:
        return json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])


def test_claims_block_is_compact_and_complete(bench):
    """the N = 1 line ends with `claims` -- every rate README.md's table quotes (value, ms, VALU fraction), the
    2^20..2^26 size table with its HBM fraction, the four stress launches (ms, value, HBM and VALU fraction), both vs_baseline
    countings and the outcome of the line's own checks -- in at most 1.5 KB, so that a record keeping only the last 2 KB of the
    line still carries all of them.  Rebuilt here from the committed line by the function bench.py itself uses."""
    d = _committed_line()
    c = bench.build_claims(d, ["x"] * 30, [])
    text = json.dumps(c)
    assert len(text) <= 1536, len(text)
    assert set(c["rates"]) == {"kernel_alone", "serial", "two_queues", "sync_call", "library_pipeline", "long_region", "runs20_single_launch",
                               "other_stepper", "q1_curve_gpu", "q3_vega_gpu", "q3_fd_vega_gpu"}
    assert c["rates"]["kernel_alone"][0] == pytest.approx(d["kernel_Mpaths_per_sec_isolated_launch"], rel=1e-5)
    assert c["rates"]["two_queues"][:2] == [pytest.approx(d["value"], rel=1e-5), pytest.approx(d["ms_per_step"], rel=1e-4)]
    assert c["rates"]["kernel_alone"][2] == pytest.approx(d["valu"]["frac"], abs=1e-4)
    # a serial call's VALU fraction follows from its duration: longer call, same work, smaller share
    assert c["rates"]["sync_call"][2] == pytest.approx(d["valu"]["frac"] * d["roofline"]["kernel_ms_avg"] / d["sync_call"]["ms_per_call"], rel=1e-3)
    assert [r[0] for r in c["size_table"]] == [20, 22, 24, 26] and len(c["size_cols"]) == 4
    for p in ("f32", "f64"):
        for k in ("zbc", "curve"):
            ms, val, hbm, valu = c["stress"][p][k]
            assert ms == pytest.approx(d["stress"][p][k]["kernel_ms"], rel=1e-4) and val == pytest.approx(2 * (1 << 26) / ms / 1e3, rel=1e-3)
            assert 0 < hbm < 1e-4 and 0.9 < valu <= 1.03
    assert c["checks_passed"] == 30 and c["checks_failed"] == [] and c["hbm_frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-2)
    assert list(c)[-2:] == ["checks_passed", "checks_failed"]
    # a failed check travels in the block, by name
    assert bench.build_claims(d, [], ["stress.f32.zbc: price 4.7 SE off"])["checks_failed"] == ["stress.f32.zbc: price 4.7 SE off"]


def test_checks_helper_and_rounding(bench):
    import numpy as np
    c = bench.Checks()
    assert c.within_se("a", 1.0, 1.0003, 1e-4) and not c.within_se("b", 1.0, 1.0005, 1e-4) and c.within_se("c", 1.0, 1.0005, 1e-4, slack=2e-4)
    assert c.rel("d", [1.0, 2.0], [1.0, 2.0 * (1 + 5e-14)], 1e-13) and not c.rel("e", [1.0, 2.0], [1.0, 2.0 * (1 + 5e-13)], 1e-13)
    assert not c.rel("shape", [1.0], [1.0, 2.0], 1e-13)
    assert c.equal("f", np.arange(3), np.arange(3)) and not c.equal("g", np.arange(3), np.arange(3) + 1)
    assert c.passed == ["a", "c", "d", "f"] and [x.split(":")[0] for x in c.failed] == ["b", "e", "shape", "g"]
    assert "SE" in c.failed[0] and "max rel diff" in c.failed[1]
    assert bench.sig(11957.7345, 6) == 11957.7 and bench.sig(2.706661e-05, 3) == 2.71e-05 and bench.sig(None) is None and bench.sig(0.0) == 0.0


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
    # without an N = 1 line the committed one-GPU constants (reference_sums.json) stand in for it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scaling.py"), many], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.splitlines()[-1])
    assert rep["n1_side"].startswith("committed constants") and rep["sums_step0_max_rel_diff"] <= 1e-13 and rep["strong_scaling"] > 7.0
    n8["sums_step0"][2] *= 1.0 + 1e-9
    json.dump({"runs": [{"n": 8, "parsed": n8}]}, open(bad, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scaling.py"), one, bad], capture_output=True, text=True)
    assert r.returncode == 1 and "did not price the N = 1 sample set" in r.stdout


def test_tools_and_labs_compile():
    """every Python tool and lab at least byte-compiles (the labs are not run by any test: they are records of experiments)"""
    import glob
    import py_compile
    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "labs", "*.py"))
    assert len(files) >= 25
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
