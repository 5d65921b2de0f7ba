#!/usr/bin/env python3
"""One GPU's share of the sharded configurations, measured on ONE GPU with the per-call all-reduce in the loop
(HWMC_BENCH_FORCE_RCCL=1: a one-rank RCCL communicator -- the collective's kernel and stream hops are real, its xGMI hop is not):

    python tools/shard_steps.py > profiles/rNN_shard_steps.txt

BASELINE configs[3] (2^24 pairs, 1000-step grid) at N = 2, 4, 8 -> 2^23, 2^22, 2^21 pairs per GPU; configs[4] (2^26 pairs,
4000-step grid, float and double) at N = 8 -> 2^23 pairs per GPU.  Each line: ms_per_step of bench.py's timed region (two queues
in flight), the same K steps without the all-reduce, one call end to end, the kernel alone -- the measured side of the
prediction table in DESIGN.md section 6 (what an N-GPU run adds is the xGMI hop and the barrier skew between processes)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    env = dict(os.environ, HWMC_BENCH_FORCE_RCCL="1", HWMC_BENCH_NO_LIVE_BOUND="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "20", "--warmup", "3", *args],
                       capture_output=True, text=True, timeout=600, env=env)
    if r.returncode != 0:
        sys.exit(r.stdout[-1500:] + r.stderr[-2000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def main():
    cases = [("configs[3] at N = 8", 21, 1000, "f32", 1 << 24, 2.761), ("configs[3] at N = 4", 22, 1000, "f32", 1 << 24, 2.761),
             ("configs[3] at N = 2", 23, 1000, "f32", 1 << 24, 2.761),
             ("configs[4] f32 at N = 8", 23, 4000, "f32", 1 << 26, None), ("configs[4] f64 at N = 8", 23, 4000, "f64", 1 << 26, None)]
    print("one GPU's share per pricing call, per-call all-reduce through a one-rank RCCL communicator; ms")
    print("%-26s %8s %12s %14s %12s %12s %10s" % ("share", "pairs", "ms_per_step", "no all-reduce", "single call", "kernel alone", "checks"))
    for name, lg, grid, prec, total, _ in cases:
        d = run("--paths-log2", str(lg), "--grid-steps", str(grid), "--precision", prec)
        mg = d["multi_gpu"]
        print("%-26s %8s %12.4f %14.4f %12.4f %12.4f %10s" % (name, "2^%d" % lg, d["ms_per_step"], mg["compute_only_ms_per_step"], mg["single_call_ms"],
                                                           d["roofline"]["kernel_ms_avg"], "%d/%d" % (d["checks"]["passed"], d["checks"]["passed"] + d["checks"]["failed"])), flush=True)


if __name__ == "__main__":
    main()
