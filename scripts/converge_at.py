#!/usr/bin/env python3
"""How many iterations does a generated workload take to stop on its own (atol = btol = conlim = 0)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen, capi
for spec in sys.argv[1:]:
    dp = devgen.generate(spec, itnlim=5000)
    s = dp.solver
    s.atol = s.btol = s.conlim = 0.0
    d_x = capi.DeviceBuffer(8 * dp.n)
    t0 = time.perf_counter()
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, dp.damp)
    print(f"{spec}: damp {dp.damp} istop {r.istop} itn {r.itn} rnorm {r.rnorm:.6e} arnorm {r.arnorm:.3e} acond {r.acond:.3e} "
          f"({time.perf_counter()-t0:.1f} s)", flush=True)
    del dp, s, d_x
