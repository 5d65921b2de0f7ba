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


def test_bench_two_rank_control_flow_rehearsal():
    """N = 2 launched exactly as the driver does (torch.distributed.run, one process per rank), in rehearsal mode: both
    ranks on the one visible GPU, all-reduce over gloo.  Not a measurement -- it pins the multi-rank control flow
    (sharding by global id, bucketed all-reduce, MAX of the elapsed time over ranks, rank 0 prints ONE line)."""
    e = dict(os.environ, HWMC_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup-ms", "50"],
                       capture_output=True, text=True, timeout=400, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(d) and d["n_gpus"] == 2 and "rehearsal" in d
    assert d["config"]["effective_paths"] == 2 * 2 * d["config"]["pairs_per_gpu"]
    # the two shards together price the option: mean of 3 runs of 2^22 samples each
    assert abs(d["price_mean"] - 0.0252555) < 5e-4
