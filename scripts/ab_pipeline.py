#!/usr/bin/env python3
"""A/B the iteration schedules on config 2 in ONE process (interleaved rounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lsqr_amd import devgen, capi

spec = sys.argv[1] if len(sys.argv) > 1 else "poisson2d:1000:1000"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dp = devgen.generate(spec, itnlim=K)
s = dp.solver
d_x = capi.DeviceBuffer(8 * dp.n)
res = {}
for rnd in range(4):
    for pipe in (0, 1):
        for gi in (10, 20, 40):
            s.set_option("pipeline", pipe); s.set_option("graph_iters", gi)
            capi.lib().lsqrhip_dev_sync()
            t0 = time.perf_counter()
            r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, dp.damp)
            dt = time.perf_counter() - t0
            assert r.itn == K
            res.setdefault((pipe, gi), []).append((dt, s.last_timing().loop_ms, r.anorm))
for k, v in sorted(res.items()):
    dts = [a[0] for a in v[1:]]; lm = [a[1] for a in v[1:]]
    print(f"pipeline={k[0]} graph_iters={k[1]:3d}: median {np.median(dts)*1e6/K:7.2f} us/it  min {min(dts)*1e6/K:7.2f}  "
          f"device loop {np.median(lm)*1e3/K:7.2f} us/it   -> {K/np.median(dts):8.0f} it/s   anorm {v[0][2]:.15e}")
