#!/usr/bin/env python3
"""Where a short solve's time goes: wall clock of solve_device, the library's own host clock (solve_ms)
and the device time between the events around the loop (loop_ms), for K = 20 and 200 at config 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lsqr_amd import capi, problems as P
from lsqr_amd.solver import lsqr_solver_ez
p = P.poisson2d(1000, 1000)
s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
s.set_option("loop_events", 1)   # timing.loop_ms is -1 without it
d_b = capi.DeviceBuffer.from_array(p.b)
d_x = capi.DeviceBuffer(8 * p.n)
for K in (20, 20, 40, 200):
    s.set_option("graph_iters", min(K, 100))
    s.itnlim = K
    s.solve_device(d_b.ptr.value, d_x.ptr.value, 0.0)
    capi.check(capi.lib().lsqrhip_dev_sync())
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        r = s.solve_device(d_b.ptr.value, d_x.ptr.value, 0.0)
        dt = time.perf_counter() - t0
        tm = s.last_timing()
        ts.append((dt * 1e6, tm.solve_ms * 1e3, tm.loop_ms * 1e3))
    best = min(ts)
    print(f"K={K:4d}: wall {best[0]:7.1f} us  library host clock {best[1]:7.1f} us  device loop (events) {best[2]:7.1f} us  = {best[2]/K:.2f} us/it", flush=True)
