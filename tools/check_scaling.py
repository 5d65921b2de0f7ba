#!/usr/bin/env python3
"""Check an N-GPU bench line against the N = 1 line of the same build (no GPU needed):

    python tools/check_scaling.py BENCH_r04.json SCALE_r04.json
    python tools/check_scaling.py line_n1.json line_n8.json
    python tools/check_scaling.py SCALE_r05.json              (no N = 1 line: the committed one-GPU constants stand in for it)

Each argument is a file holding bench.py's JSON line, a driver record with the line under "parsed", or a driver scaling
record holding several runs (a list, or a dict of per-N records): every N > 1 line found is checked against the N = 1 line
(hwmc.parallel.check_scaling_lines: every rank took part, the SAME sample set was priced -- sums of the first timed step equal
to 1e-13 --, value = paths / time) and its report is printed: strong_scaling, compute_scaling, exposed all-reduce, launch skew,
RCCL against host-sum call latency, and a verdict in words ("scales" / "compute-bound shortfall" / "collective exposed")."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def lines_in(obj):
    """every bench line (a dict with metric / n_gpus / ms_per_step) found anywhere inside a parsed JSON document"""
    if isinstance(obj, dict):
        if {"metric", "n_gpus", "ms_per_step"} <= set(obj):
            yield obj
            return
        for v in obj.values():
            yield from lines_in(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from lines_in(v)


def load(path):
    text = open(path).read()
    try:
        return list(lines_in(json.loads(text)))
    except ValueError:
        return [d for d in (json.loads(l) for l in text.splitlines() if l.startswith("{")) if "metric" in d]


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    import hwmc
    found = [d for p in sys.argv[1:] for d in load(p)]
    ones = [d for d in found if d["n_gpus"] == 1 and d.get("strong_scaling_reference")]
    many = [d for d in found if d["n_gpus"] > 1]
    if not many:
        print("need at least one N > 1 line (found %d N = 1 / %d N > 1)" % (len(ones), len(many)))
        return 2
    rc = 0
    for d in sorted(many, key=lambda x: x["n_gpus"]):
        try:
            one = ones[0] if ones else hwmc.parallel.committed_one_gpu_line(d)[0]
            if one is None or not one["strong_scaling_reference"].get("ms_per_step"):
                print("N = %d: no N = 1 line given and no committed one-GPU constant for its configuration" % d["n_gpus"])
                rc = 1
                continue
            rep = hwmc.parallel.check_scaling_lines(one, d)
            rep["n1_side"] = "the N = 1 line given" if ones else "committed constants (reference_sums.json)"
            print(json.dumps(rep))
        except AssertionError as exc:
            print("N = %d: FAILED: %s" % (d["n_gpus"], exc))
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
