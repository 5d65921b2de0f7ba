import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import hwmc, hw_oracle
N = 1 << 20
p = hwmc.default_params("readme")
o = hw_oracle.Oracle("readme")
with hwmc.HullWhiteEngine(p, device=0) as e:
    for sg in (0.099, 0.101):
        got = e.simulate_zcb(N, 99, sigma=sg)
        want = o.zcb_curve(N, 99, sigma=sg)
        rel = np.abs(got / want - 1.0)
        print("sigma", sg, "max rel diff", rel.max(), "at", int(rel.argmax()), "P50 rel", got[50] / want[50] - 1.0, "P100 rel", got[100] / want[100] - 1.0)
        Pg, fg = e.compute_average_and_forward(got, 2 * N)
        Po, fo = o.average_and_forward(want, 2 * N)
        print("   f50 engine %.9f oracle %.9f ; P50 %.9f %.9f ; P100 %.9f %.9f" % (fg[50], fo[50], Pg[50], Po[50], Pg[100], Po[100]))
    os.environ["HWMC_CURVE_GENERIC"] = "1"
with hwmc.HullWhiteEngine(p, device=0) as g:
    got = g.simulate_zcb(N, 99, sigma=0.099)
    want = o.zcb_curve(N, 99, sigma=0.099)
    print("generic kernel max rel diff", np.abs(got / want - 1.0).max())
