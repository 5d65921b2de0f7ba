#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into the files kept under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r01 r01 [kernel-name-substring]

Writes profiles/<tag>_bench_kernel_stats.csv (rocprofv3 --stats table, verbatim), and
profiles/<tag>_pmc_summary.json: per-launch means of every counter for the dominant kernel, the HBM traffic
corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, both in KB), and -- from the kernel
trace -- the launch durations together with the time the kernel actually occupied the chip (union of the
dispatch intervals / launches), which is what matters when two streams keep two launches in flight."""
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    # the float ZBC kernel with two pairs per thread: what the engine launches for the default bench (2^20 pairs: effective_ilp)
    want = sys.argv[3] if len(sys.argv) > 3 else "zbc_superposed_kernel<float, 1, 2>"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {"command": open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else None,
           "kernel": want, "passes": {}}
    stats = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(root, "profiles", "%s_bench_kernel_stats.csv" % tag))
    stats1 = newest(os.path.join(src, "trace_s1", "**", "*kernel_stats.csv"))
    if stats1:
        shutil.copy(stats1, os.path.join(root, "profiles", "%s_bench_streams1_kernel_stats.csv" % tag))
        with open(stats1) as fh:
            for r in csv.DictReader(fh):
                if want in r["Name"]:
                    out["single_stream_run"] = {"command": (out["command"] or "") + " --streams 1", "launches": int(r["Calls"]),
                                                "avg_duration_us": float(r["AverageNs"]) / 1e3, "min_duration_us": float(r["MinNs"]) / 1e3}
    trace = newest(os.path.join(src, "trace", "**", "*kernel_trace.csv"))
    if trace:
        iv = []
        with open(trace) as fh:
            for r in csv.DictReader(fh):
                if want in r["Kernel_Name"]:
                    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
        iv.sort()
        dur = [b - a for a, b, _ in iv]
        union, cur_a, cur_b = 0, None, None
        for a, b, _ in iv:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    union += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            union += cur_b - cur_a
        overlapped = sum(1 for i in range(1, len(iv)) if iv[i][0] < max(x[1] for x in iv[max(0, i - 4):i]))
        # the same union restricted to the PIPELINED stretches of the run (launches that share the chip with a launch of
        # another queue): the process also runs single-stream passes (serial rate, isolated kernel), whose launches would
        # otherwise pull the figure towards the isolated duration
        pip = []
        for i, (a, b, q) in enumerate(iv):
            lo, hi = max(0, i - 3), min(len(iv), i + 4)
            if any(j != i and iv[j][2] != q and iv[j][0] < b and iv[j][1] > a for j in range(lo, hi)):
                pip.append((a, b))
        pun, ca, cb = 0, None, None
        for a, b in pip:
            if cb is None or a > cb:
                if cb is not None:
                    pun += cb - ca
                ca, cb = a, b
            else:
                cb = max(cb, b)
        if cb is not None:
            pun += cb - ca
        out["kernel_trace"] = {
            "launches": len(iv), "queues": sorted({q for _, _, q in iv}),
            "avg_duration_us": sum(dur) / len(dur) / 1e3, "min_duration_us": min(dur) / 1e3, "max_duration_us": max(dur) / 1e3,
            "launches_starting_before_the_previous_one_ended": overlapped,
            "chip_time_per_launch_us": union / len(iv) / 1e3,
            "pipelined_launches": len(pip),
            "chip_time_per_pipelined_launch_us": (pun / len(pip) / 1e3) if pip else None,
            "note": "chip_time_per_launch = union of the dispatch intervals / launches: with two queues in flight a "
                    "dispatch lasts about twice as long as it occupies the chip for",
        }
    # the bench JSON line printed BY THE TRACED PROCESS itself (default command and --streams 1): laid beside the
    # kernel-trace figures of the same process so that ms_per_step, kernel_ms_avg and the trace can be compared
    for key, log, dst in (("traced_default_run", "trace.log", "%s_bench_traced_default.json"), ("traced_streams1_run", "trace_s1.log", "%s_bench_traced_streams1.json")):
        path = os.path.join(src, log)
        if not os.path.exists(path):
            continue
        lines = [l for l in open(path, errors="replace").read().splitlines() if l.startswith('{"metric"')]
        if not lines:
            continue
        with open(os.path.join(root, "profiles", dst % tag), "w") as fh:
            fh.write(lines[-1] + "\n")
        d = json.loads(lines[-1])
        out[key] = {"file": "profiles/" + dst % tag, "value": d["value"], "ms_per_step": d["ms_per_step"],
                    "value_spread": d.get("value_spread"),
                    "kernel_ms_avg_isolated_hip_events": d["roofline"]["kernel_ms_avg"],
                    "kernel_ms_per_step_in_timed_region": d["roofline"].get("kernel_ms_per_step_in_timed_region"),
                    "serial_ms_per_step": d["serial"]["ms_per_step"],
                    "note": "numbers of the process rocprofv3 was tracing (tracing itself costs a few per cent)"}
    if "traced_default_run" in out and "kernel_trace" in out:
        out["consistency"] = {
            "default_run": {"ms_per_step_bench": out["traced_default_run"]["ms_per_step"],
                            "chip_time_per_launch_ms_trace": out["kernel_trace"]["chip_time_per_launch_us"] / 1e3,
                            "chip_time_per_pipelined_launch_ms_trace": (out["kernel_trace"]["chip_time_per_pipelined_launch_us"] or 0) / 1e3,
                            "note": "the trace covers warm-up, stream selection and side measurements too; the union per launch is the "
                                    "comparable figure (<= ms_per_step when the timed region is representative)"}}
        if "single_stream_run" in out and "traced_streams1_run" in out:
            out["consistency"]["streams1_run"] = {
                "kernel_ms_avg_bench_hip_events": out["traced_streams1_run"]["kernel_ms_avg_isolated_hip_events"],
                "kernel_avg_ms_trace": out["single_stream_run"]["avg_duration_us"] / 1e3,
                "serial_ms_per_step_bench": out["traced_streams1_run"]["serial_ms_per_step"]}
    for pas in ("pmc_sq", "pmc_fetch", "pmc_write"):
        f = newest(os.path.join(src, pas, "**", "*counter_collection.csv"))
        if not f:
            continue
        acc = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if want in r["Kernel_Name"]:
                    a = acc.setdefault(r["Counter_Name"], [0, 0.0])
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
        out["passes"][pas] = {k: {"launches": v[0], "mean_per_launch": v[1] / v[0]} for k, v in sorted(acc.items())}
    try:
        fk = out["passes"]["pmc_fetch"]["FETCH_SIZE"]["mean_per_launch"]
        wk = out["passes"]["pmc_write"]["WRITE_SIZE"]["mean_per_launch"]
        out["hbm_traffic_bytes_per_launch"] = {
            "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "corrected_bytes": (2.0 * fk + wk) * 1024.0,
            "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as read; both in KB"}
    except KeyError:
        pass
    sq = out["passes"].get("pmc_sq", {})
    if "SQ_INSTS_VALU" in sq and "SQ_WAVES" in sq:
        out["valu_instructions_per_wave"] = sq["SQ_INSTS_VALU"]["mean_per_launch"] / sq["SQ_WAVES"]["mean_per_launch"]
    dst = os.path.join(root, "profiles", "%s_pmc_summary.json" % tag)
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out.get("kernel_trace"), indent=1))
    print(json.dumps(out.get("hbm_traffic_bytes_per_launch"), indent=1))


if __name__ == "__main__":
    main()
