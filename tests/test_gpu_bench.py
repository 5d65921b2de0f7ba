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


def _bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--spinup-ms", "50", *args],
                       capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("args,env", [
    (["--gpus", "1", "--steps", "4", "--warmup", "1"], None),
    (["--steps", "1", "--warmup", "0"], None),
    (["--steps", "3", "--streams", "1"], None),
    (["--steps", "3", "--streams", "3", "--paths-log2", "18", "--grid-steps", "4000"], None),
    (["--steps", "3", "--precision", "f64"], None),
    (["--steps", "3", "--stepper", "reference", "--profile", "readme"], None),
    (["--steps", "5", "--warmup", "0", "--allreduce-every", "2"], {"HWMC_BENCH_FORCE_RCCL": "1"}),
    (["--steps", "4"], {"HWMC_BENCH_FORCE_RCCL": "1"}),
])
def test_bench_flag_combinations(args, env):
    d = _bench(*args, env=env)
    assert CONTRACT_KEYS <= set(d), sorted(CONTRACT_KEYS - set(d))
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["unit"] == "M paths/s"
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert 0.0 < d["price_mean"] < 0.1 and d["config"]["workload"].startswith("Q2b option_pricing")
    assert d["steps"] == int(args[args.index("--steps") + 1])
