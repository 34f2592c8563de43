#!/usr/bin/env python3
"""Look-ahead poll (option poll_ahead) vs strict launch-wait-check: wall-clock per iteration at
config 2 for fixed iteration counts, and the cost of the no-op tail on a converging solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lsqr_amd import devgen, capi

dp = devgen.generate("poisson2d:1000:1000", itnlim=400)
s = dp.solver
d_x = capi.DeviceBuffer(8 * dp.n)
for K in (400, 2000):
    s.itnlim = K
    for gi in (20, 50, 100):
        for pa in (0, 1, 0, 1):
            s.set_option("graph_iters", gi); s.set_option("poll_ahead", pa)
            ts = []
            for rep in range(4):
                capi.lib().lsqrhip_dev_sync()
                t0 = time.perf_counter()
                r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 0.0)
                ts.append(time.perf_counter() - t0)
            assert r.itn == K
            print(f"K={K:5d} graph_iters={gi:4d} poll_ahead={pa}: {1e6*min(ts[1:])/K:7.3f} us/it wall "
                  f"({K/min(ts[1:]):8.0f} it/s)  anorm {r.anorm:.15e}", flush=True)
# a solve that converges (stops mid-batch): total wall time
dq = devgen.generate("poisson2d:300:300", itnlim=100000)
q = dq.solver
q.atol = q.btol = 1e-8
d_y = capi.DeviceBuffer(8 * dq.n)
for gi in (20, 50):
    for pa in (0, 1, 0, 1):
        q.set_option("graph_iters", gi); q.set_option("poll_ahead", pa)
        ts = []
        for rep in range(4):
            capi.lib().lsqrhip_dev_sync()
            t0 = time.perf_counter()
            r = q.solve_device(dq.d_b.ptr.value, d_y.ptr.value, 0.0)
            ts.append(time.perf_counter() - t0)
        print(f"converging 300x300 graph_iters={gi:3d} poll_ahead={pa}: itn {r.itn} istop {r.istop} "
              f"{1e3*min(ts[1:]):8.3f} ms  rnorm {r.rnorm:.15e}", flush=True)
