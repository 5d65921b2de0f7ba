#!/usr/bin/env python3
"""Write <package>/reference_sums.json: the moment sums of whole pricing calls on ONE GPU that an N-GPU run must reproduce.

    python tools/make_reference_sums.py [--out FILE] [--check]

Entries (seed 1234 = the first timed step of bench.py, 2_option:229; theta profile `head`):
  * ZBC, 1000-step grid, float, 2^20 and 2^24 pairs            -- BASELINE configs[1] and configs[3]
  * ZBC, 4000-step grid, 2^26 pairs, float and double          -- BASELINE configs[4] (bench.py --total-log2 26 --grid-steps 4000)
  * curve (simulate_zcb), 4000-step grid, 2^26 pairs, float and double: P_sum[101]
Runs on an MI355X (~1 s of GPU time).  --check compares a fresh run with the committed file (1e-13) instead of writing it:
the sums are deterministic for a given library build -- the driver's box reproduced the builder's to the last bit in round 4."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import hwmc
    par = hwmc.parallel
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=par.REFERENCE_SUMS_FILE)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    have = par.load_reference_sums()
    out = {"_about": have.get("_about", "")}
    seed = 1234
    for prec, grid, sizes in (("f32", 1000, (20, 24)), ("f32", 4000, (26,)), ("f64", 4000, (26,))):
        p = hwmc.default_params("head", n_steps=grid, precision=prec)
        P, f = hwmc.analytic_curve(p)
        with hwmc.HullWhiteEngine(p, device=0) as e:
            e.load_market_data(P, f)
            for lg in sizes:
                out[par.reference_key("zbc", "head", prec, "superposed", grid, lg, seed)] = [float(x) for x in e.simulate_ZBC_control_variate(1 << lg, seed)]
                if grid == 4000:
                    out[par.reference_key("zcb", "head", prec, "blocked", grid, lg, seed)] = [float(x) for x in e.simulate_zcb(1 << lg, seed)]
    if a.check:
        worst = 0.0
        for k, v in out.items():
            if k.startswith("_"):
                continue
            assert k in have, "no committed entry %s" % k
            d = max(abs(x - y) / max(abs(y), 1e-300) for x, y in zip(v, have[k]))
            worst = max(worst, d)
            print("%-48s max rel diff %.2e" % (k, d))
        print("worst %.2e" % worst)
        return 0 if worst <= 1e-13 else 1
    out.update({k: v for k, v in have.items() if k.startswith("_") and k != "_about"})      # committed times and predictions stay
    with open(a.out, "w") as fh:
        fh.write("{\n" + ",\n".join(" %s: %s" % (json.dumps(k), json.dumps(v)) for k, v in out.items()) + "\n}\n")
    print("wrote %s (%d sums)" % (a.out, sum(1 for k in out if not k.startswith("_"))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
