#!/usr/bin/env python3
"""bench.py at the batch sizes the north star names (2^20 .. 2^26 pairs per step, one GPU) and the stress grid, one line each.
    python tools/bench_sizes.py > profiles/rNN_bench_size_table.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--regions", "3", *args],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        sys.exit(r.stderr[-2000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def main():
    for k in (20, 22, 24, 26):
        d = run("--steps", "10", "--warmup", "2", "--paths-log2", str(k))
        print("2^%d pairs/step, 1000-step grid, f32: value %.0f M paths/s (regions %.0f..%.0f), serial %.0f, kernel alone %.0f, "
              "ms/step %.4f, VALU frac isolated %.3f" % (k, d["value"], d["value_spread"]["min"], d["value_spread"]["max"], d["serial"]["value"],
                                                          d["kernel_Mpaths_per_sec_isolated_launch"], d["ms_per_step"],
                                                          d["valu"]["frac_of_nominal_clock_peak_isolated_launch"]), flush=True)
    for prec in ("f32", "f64"):
        d = run("--steps", "3", "--warmup", "1", "--paths-log2", "26", "--grid-steps", "4000", "--precision", prec)
        print("stress 2^26 pairs x 4000-step grid, %s: %.2f ms/step, %.0f M paths/s, VALU frac isolated %.3f"
              % (prec, d["ms_per_step"], d["value"], d["valu"]["frac_of_nominal_clock_peak_isolated_launch"]), flush=True)


if __name__ == "__main__":
    main()
