#!/usr/bin/env python3
"""Device-loop time per iteration vs graph batch size (cost of the per-batch host poll)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen, capi
K = 800
dp = devgen.generate("poisson2d:1000:1000", itnlim=K)
s = dp.solver
s.set_option("loop_events", 1)   # timing.loop_ms is -1 without it
d_x = capi.DeviceBuffer(8 * dp.n)
for gi in (20, 50, 100, 200, 400, 800, 50, 800):
    s.set_option("graph_iters", gi)
    best = 1e9
    for k in range(4):
        r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 0.0)
        best = min(best, s.last_timing().loop_ms)
    print(f"graph_iters {gi:4d}: {1e3*best/K:7.3f} us/iter (itn {r.itn})", flush=True)
