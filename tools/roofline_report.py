#!/usr/bin/env python3
"""The roofline figures of a bench line as one table (no GPU needed):

    python tools/roofline_report.py [profiles/r05_bench_n1.json]

One row per measured launch: time, M paths/s, algorithmic HBM bytes and the fraction of the 8 TB/s HBM roof they amount to,
achieved shader cycles per Philox pair-block per SIMD (launch time x in-kernel clock x SIMDs / pair-blocks), the measured
issue bound of THAT launch's loop (tools/mix_bound.hip) and their ratio."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))[-1]
    d = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    rows = []

    def row(name, ms, rate, hbm_frac, valu, loop=None):
        rows.append((name, ms, rate, hbm_frac, (valu or {}).get("achieved_cycles_per_block_isolated_launch"),
                     (valu or {}).get("bound_cycles_per_block"), (valu or {}).get("frac"), loop or (valu or {}).get("loop") or ""))

    r, v = d["roofline"], d["valu"]
    row("ZBC 2^20 pairs, one launch alone (the metric's kernel)", r["kernel_ms_avg"], d["kernel_Mpaths_per_sec_isolated_launch"], r["frac"], v, "zbc_" + d["dtype"])
    row("ZBC 2^20 pairs inside the two-queue timed region", d["ms_per_step"], d["value"], None,
        {"achieved_cycles_per_block_isolated_launch": v["achieved_cycles_per_block_timed_region"], "bound_cycles_per_block": v["bound_cycles_per_block"],
         "frac": v["frac_timed_region"]}, "zbc_" + d["dtype"])
    for t in (d.get("size_table") or {}).get("rows", []):
        row("ZBC 2^%d pairs, one launch alone" % t["pairs_log2"], t["kernel_ms"], t["value"], None,
            {"frac": t["valu_frac"], "bound_cycles_per_block": v["bound_cycles_per_block"],
             "achieved_cycles_per_block_isolated_launch": v["bound_cycles_per_block"] / t["valu_frac"] if t["valu_frac"] else None}, "zbc_" + d["dtype"])
    if "q3_vega_gpu" in d:
        row("pathwise Vega, 2^20 paths", d["q3_vega_gpu"]["kernel_ms"], d["q3_vega_gpu"]["value"], None, d["q3_vega_gpu"].get("valu"), "zbc_" + d["dtype"])
    if "q3_fd_vega_gpu" in d:
        row("FD Vega: 3 scenarios in one pass, 2^20 pairs", d["q3_fd_vega_gpu"]["kernel_ms"], d["q3_fd_vega_gpu"]["value"], None, d["q3_fd_vega_gpu"]["valu"])
    if "q1_curve_gpu" in d:
        row("curve, 2^20 pairs x 1000 steps x 100 maturities", d["q1_curve_gpu"]["kernel_ms"], d["q1_curve_gpu"]["value"], None, d["q1_curve_gpu"].get("valu"))
    for prec in ("f32", "f64"):
        for k in ("zbc", "curve"):
            x = (d.get("stress") or {}).get(prec, {}).get(k)
            if x:
                row("stress 2^26 x 4000-step grid, %s, %s" % (k, prec), x["kernel_ms"], x["value"], x["hbm"]["frac"], x["valu"])
    print("bench line: %s   device: %s" % (os.path.relpath(path, ROOT), d.get("device")))
    print("%-56s %10s %10s %10s %9s %8s %6s  %s" % ("launch", "ms", "M paths/s", "HBM frac", "achieved", "bound", "frac", "loop (twin)"))
    for name, ms, rate, hf, ach, bnd, fr, loop in rows:
        print("%-56s %10.4f %10.0f %10s %9s %8s %6s  %s" % (name, ms, rate, ("%.1e" % hf) if hf is not None else "", ("%.1f" % ach) if ach else "",
                                                          ("%.1f" % bnd) if bnd else "", ("%.3f" % fr) if fr else "", loop))
    cb = d.get("cpu_baseline")
    if cb:
        print("CPU baseline (oracle, %d cores): ZBC loop %.2f M paths/s (1 thread %.3f); Q1 curve loop %.2f M paths/s (1 thread %.3f)"
              % (cb["cores"], cb["value"], cb["single_thread_value"], cb["q1_curve"]["value"], cb["q1_curve"]["single_thread_value"]))
    print("HBM: algorithmic bytes per launch / launch time / 8 TB/s (stateless Philox: the path moves ~0 bytes); "
          "bound: cycles per pair-block per SIMD of the loop's dependency-free twin (profiles/rNN_mix_bound.txt)")


if __name__ == "__main__":
    main()
