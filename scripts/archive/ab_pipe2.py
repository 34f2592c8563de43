#!/usr/bin/env python3
"""Device-loop time per iteration at config 2 for the launch schedules (pipeline 0/1/2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen, capi
K = 800
dp = devgen.generate(sys.argv[1] if len(sys.argv) > 1 else "poisson2d:1000:1000", itnlim=K)
s = dp.solver
s.set_option("loop_events", 1)   # timing.loop_ms is -1 without it
d_x = capi.DeviceBuffer(8 * dp.n)
s.set_option("graph_iters", 100)
for pipe in (2, 1, 0, 2, 1):
    s.set_option("pipeline", pipe)
    best = 1e9
    for k in range(4):
        r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 0.0)
        best = min(best, s.last_timing().loop_ms)
    print(f"pipeline {pipe}: {1e3*best/K:7.3f} us/iter (itn {r.itn})", flush=True)
for w in (1, 2, 3):
    print("kernel", w, f"{1e3*s.bench_kernel(w, 400):.3f} us")
