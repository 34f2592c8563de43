#!/usr/bin/env python3
"""One schedule only (for rocprofv3 --kernel-trace): usage one_pipe.py PIPELINE [SPEC]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen, capi
K = 800
pipe = int(sys.argv[1])
dp = devgen.generate(sys.argv[2] if len(sys.argv) > 2 else "poisson2d:1000:1000", itnlim=K)
s = dp.solver
s.set_option("loop_events", 1)   # timing.loop_ms is -1 without it
d_x = capi.DeviceBuffer(8 * dp.n)
s.set_option("graph_iters", 100)
s.set_option("pipeline", pipe)
for k in range(3):
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 0.0)
    print(f"pipeline {pipe}: {1e3*s.last_timing().loop_ms/K:7.3f} us/iter (itn {r.itn})", flush=True)
