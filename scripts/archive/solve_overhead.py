#!/usr/bin/env python3
"""Fixed cost of one solve at config 2: wall time of solve_device for several iteration counts and
graph batch sizes (fit: fixed + K * per-iteration)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lsqr_amd import devgen, capi
dp = devgen.generate("poisson2d:1000:1000", itnlim=10)
s = dp.solver
d_x = capi.DeviceBuffer(8 * dp.n)
for gi in (10, 32, 64, 100):
    s.set_option("graph_iters", gi)
    row = []
    for K in (2, 10, 50, 100, 200, 400):
        s.itnlim = K
        ts = []
        for rep in range(5):
            capi.lib().lsqrhip_dev_sync()
            t0 = time.perf_counter()
            r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 0.0)
            ts.append(time.perf_counter() - t0)
        row.append((K, 1e6 * min(ts[1:])))
    (k1, t1), (k2, t2) = row[-2], row[-1]
    per = (t2 - t1) / (k2 - k1)
    print(f"graph_iters {gi:4d}: " + "  ".join(f"K={k}: {t:7.0f} us" for k, t in row) + f"   per-iteration {per:.2f} us, fixed {row[-1][1] - per * row[-1][0]:.0f} us", flush=True)
