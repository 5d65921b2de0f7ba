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
              "ms/step %.4f, VALU bound/achieved: isolated launch %.3f (%.1f cycles per block), timed region %.3f; clock %.0f MHz"
              % (k, d["value"], d["value_spread"]["min"], d["value_spread"]["max"], d["serial"]["value"],
                 d["kernel_Mpaths_per_sec_isolated_launch"], d["ms_per_step"], d["valu"]["frac"],
                 d["valu"]["achieved_cycles_per_block_isolated_launch"], d["valu"]["frac_timed_region"], d["valu"]["clock_mhz_in_kernel"] or 0), flush=True)
    for prec in ("f32", "f64"):
        d = run("--steps", "3", "--warmup", "1", "--paths-log2", "26", "--grid-steps", "4000", "--precision", prec)
        v = d["valu"]
        print("stress 2^26 pairs x 4000-step grid, %s: %.2f ms/step, %.0f M paths/s, achieved %.1f cycles per block (isolated launch), bound/achieved %s; clock %.0f MHz"
              % (prec, d["ms_per_step"], d["value"], v["achieved_cycles_per_block_isolated_launch"],
                 ("%.3f" % v["frac"]) if v["frac"] else "n/a (no bound for this loop in the probe result)", v["clock_mhz_in_kernel"] or 0), flush=True)


if __name__ == "__main__":
    main()
