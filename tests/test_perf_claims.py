"""The absolute-throughput claims README.md / DESIGN.md make (north star: >= 10x the reference's V100 figure of 1064 M
paths/s for the ZBC option path), pinned in two ways that cannot make the correctness suite flaky:

  * without a GPU: against the bench line COMMITTED with the newest round's profiles (profiles/rNN_bench_n1.json, the
    output of the driver's command on an MI355X) -- the claim and its evidence cannot drift apart;
  * on an MI355X, `-m perf` (NOT part of `-m gpu`): the same bars measured live through the C ABI.
"""
import glob
import json
import os
import re
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAR = 10.0 * 1064.0            # M paths/s: 10x README.md:44,162 (one V100)


def _newest_bench_line():
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")):
        m = re.match(r"r(\d+)_bench_n1\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    assert best, "no committed bench line under profiles/"
    with open(best[1]) as fh:
        return best[0], json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])


def test_committed_bench_line_carries_the_claims():
    rnd, d = _newest_bench_line()
    assert d["metric"].startswith("M paths/sec (ZBC option") and d["n_gpus"] == 1 and d["config"]["pairs_per_step"] == 1 << 20
    # like for like with the reference's single timed launch first, the pipelined figure second
    assert d["kernel_Mpaths_per_sec_isolated_launch"] >= BAR            # the path kernel alone on the chip
    assert d["serial"]["value"] >= BAR                                   # K calls strictly one after another, wall clock
    assert d["value"] >= BAR and d["value"] >= d["serial"]["value"]      # two queues in flight
    assert d["vs_baseline"] == pytest.approx(d["value"] / 1064.0, rel=1e-3)
    if rnd >= 3:
        v = d["valu"]
        assert 0.0 < v["frac"] <= 1.03 and 0.0 < v["frac_timed_region"] <= 1.03    # a measured reference rate (3 % tolerance: both sides are live measurements)
        assert d["q3_vega_gpu"]["vs_v100_509"] >= 10.0 and d["q1_curve_gpu"]["vs_v100_391"] >= 10.0
        assert d["timed_region_s"] >= 0.5


@pytest.mark.perf
def test_live_pipelined_and_synchronous_rates(hwmc_mod):
    import numpy as np
    p = hwmc_mod.default_params("head")
    P, f = hwmc_mod.analytic_curve(p)
    n, seeds = 1 << 20, [7 + k for k in range(40)]
    with hwmc_mod.HullWhitePipeline(p, device=0, queues=2) as pipe, hwmc_mod.HullWhiteEngine(p, device=0) as e:
        pipe.load_market_data(P, f)
        e.load_market_data(P, f)
        pipe.tune(n)
        t_end = time.perf_counter() + 0.5
        while time.perf_counter() < t_end:
            pipe.simulate_ZBC_batch(n, seeds)
        piped = min(pipe.simulate_ZBC_batch(n, seeds, return_ms=True)[1] for _ in range(5)) / len(seeds)
        assert 2 * n / piped / 1e3 >= BAR, piped                         # measured 12 100+
        ms = []
        for s in seeds:
            ms.append(e.simulate_ZBC_control_variate(n, s, return_ms=True)[1])
        assert 2 * n / float(np.median(ms)) / 1e3 >= BAR, np.median(ms)  # hipEvent time of a synchronous call's launches: 11 000+
