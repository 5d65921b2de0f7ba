import sys, time, numpy as np
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hwmc
p = hwmc.default_params("head")
e = hwmc.HullWhiteEngine(p, device=0)
e.load_market_data(*hwmc.analytic_curve(p))
out = torch.zeros((400, 5), dtype=torch.float64, device="cuda")
st = torch.cuda.Stream()
for n in (1 << 10, 1 << 20):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(400):
            e.enqueue_ZBC(out[i].data_ptr(), n, 1 + i, stream=st.cuda_stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n}: host enqueue {1e6*(t1-t0)/400:.1f} us/step, total {1e6*(t2-t0)/400:.1f} us/step")
