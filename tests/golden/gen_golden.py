#!/usr/bin/env python3
"""Writes tests/golden/oracle_vectors.json from the CPU oracle (oracle/libhw_oracle.so).

The reference holds no golden vectors, fixed seeds or unit tests and its
CUDA cannot be built here, so these vectors are outputs of this repository's own restatement of
the reference's algorithm on this repository's own RNG stream; what pins the oracle itself is
listed in oracle/hw_oracle.h.  Regenerate with:  python tests/golden/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hw_oracle as ho  # noqa: E402
import hwmc  # noqa: E402  (host-only: analytic curve)

SEEDS = [1234, 20240613]
K = float(np.exp(np.float32(-0.1)))


def main():
    out = {"K": K, "cases": []}
    for profile in ("head", "readme"):
        p = hwmc.default_params(profile)
        P, f = hwmc.analytic_curve(p)
        for prec in ("f32", "f64"):
            o = ho.Oracle(profile, precision=prec)
            for seed in SEEDS:
                for log2n in (10, 14):
                    n = 1 << log2n
                    case = {"profile": profile, "precision": prec, "seed": seed, "n": n,
                            "path_offset": 0}
                    case["zbc_sums"] = o.zbc_moments(P, f, n, seed).tolist()
                    case["vega_sum"] = o.vega_pathwise(P, f, n, seed)
                    if log2n == 10:
                        case["zcb_P_sum"] = o.zcb_curve(n, seed).tolist()
                        eps = 1e-3
                        s0 = o.sigma
                        fd = []
                        for sg in (s0 - eps, s0 + eps):
                            d = o.shifted_drift(sg, s0)
                            fd.append(o.zbc_moments(P, f, n, seed, sigma=sg, drift=d).tolist())
                        case["zbc_sums_fd"] = fd
                    out["cases"].append(case)
            # sharding / ragged sizes: path_offset and a size that is not a block multiple
            case = {"profile": profile, "precision": prec, "seed": 777, "n": 1000, "path_offset": 123456789012}
            case["zbc_sums"] = o.zbc_moments(P, f, 1000, 777, path_offset=123456789012).tolist()
            case["vega_sum"] = o.vega_pathwise(P, f, 1000, 777, path_offset=123456789012)
            out["cases"].append(case)
        o = ho.Oracle(profile)
        d, sd = o.tables()
        o64 = ho.Oracle(profile, precision="f64")
        d64, sd64 = o64.tables()
        out["tables_" + profile] = {
            "idx": [0, 1, 499, 500, 999],
            "drift_f32": [float(d[i]) for i in (0, 1, 499, 500, 999)],
            "sigma_drift_f32": [float(sd[i]) for i in (0, 1, 499, 500, 999)],
            "drift_f64": [float(d64[i]) for i in (0, 1, 499, 500, 999)],
            "sigma_drift_f64": [float(sd64[i]) for i in (0, 1, 499, 500, 999)],
            "constants_f32": [float(x) for x in o.constants()],
            "constants_f64": [float(x) for x in o64.constants()],
        }
        out["curve_" + profile] = {"P": [float(x) for x in P], "f": [float(x) for x in f]}
        out["sample_paths_" + profile] = {
            "seed": 4242, "n": 3,
            "r_at": [0, 1, 2, 3, 4, 5, 250, 500, 999, 1000],
            "r": o.sample_paths(3, 4242)[:, [0, 1, 2, 3, 4, 5, 250, 500, 999, 1000]].astype(float).tolist()}
    o = ho.Oracle("head")
    out["normals"] = [{"seed": s, "subsequence": q, "block": b,
                       "g": [float(x) for x in o.normal4(s, q, b)], "w": o.uint4(s, q, b)}
                      for s in SEEDS for q in (0, 1, 2, 3) for b in (0, 1, 124)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
