#!/usr/bin/env python3
"""One device-resident solve (for profiling): one_solve.py SPEC ITERS PIPELINE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen, capi
spec, K, pipe = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dp = devgen.generate(spec, itnlim=K)
s = dp.solver
s.set_option("loop_events", 1)   # timing.loop_ms is -1 without it
s.set_option("pipeline", pipe); s.set_option("graph_iters", 20)
d_x = capi.DeviceBuffer(8 * dp.n)
for _ in range(2):
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, dp.damp)
print("itn", r.itn, "loop_ms", s.last_timing().loop_ms)
