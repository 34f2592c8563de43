#!/usr/bin/env python3
"""One-off check of the 64-bit row-pointer build at real scale: 2.2e9 nonzeros (> 2^31) generated
in HBM, acheck's adjoint identity (A and A' agree), a short solve, timing of both products."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen, capi
spec = sys.argv[1] if len(sys.argv) > 1 else "random:22000000:1000000:100"
t0 = time.perf_counter()
dp = devgen.generate(spec, itnlim=6)
s = dp.solver
s.set_option("loop_events", 1)   # timing.loop_ms is -1 without it
info = s.info()
print(f"{spec}: nnz {dp.nnz} ({dp.nnz / 2**31:.2f} x 2^31)  rowptr bytes {info['rowptr_bytes']}  panels {info['panels']}/{info['panels_t']}"
      f"  build {s.build_seconds:.2f} s  total {time.perf_counter() - t0:.1f} s", flush=True)
inform, err = s.acheck()
print(f"acheck inform {inform} relative error {err:.2e}", flush=True)
d_x = capi.DeviceBuffer(8 * dp.n)
r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, dp.damp)
tm = s.last_timing()
print(f"solve: istop {r.istop} itn {r.itn} anorm {r.anorm:.6e} rnorm {r.rnorm:.6e}  {tm.loop_ms / max(r.itn, 1):.1f} ms/iteration", flush=True)
t = [s.bench_kernel(w, 3) for w in (1, 2)]
print(f"spmv1 {t[0]:.2f} ms ({dp.nnz / t[0] / 1e6:.0f} G gathers/s)  spmv2 {t[1]:.2f} ms", flush=True)
