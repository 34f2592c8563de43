#!/usr/bin/env python3
"""Why is bench.py --steps 20 slower than a bare loop of 20-iteration solves?  Same build_workload / timed_solve,
repeated, with the pieces timed apart."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lsqr_amd import capi
import torch
s, d_b, facts, host = bench.build_workload(bench.HEADLINE, None, itnlim=20)
d_x = capi.DeviceBuffer(8 * facts["n"])
s.set_option("graph_iters", 20)
s.atol = s.btol = s.conlim = 0.0
s.itnlim = 5
s.solve_device(d_b.ptr.value, d_x.ptr.value, facts["damp"])
for rep in range(6):
    s.set_option("loop_events", 1)
    dt, r, restarts = bench.timed_solve(s, d_b, d_x, facts["damp"], 20)
    loop_ms = s.last_timing().loop_ms
    print(f"timed_solve: {dt*1e6:7.1f} us  device loop {loop_ms*1e3:7.1f} us", flush=True)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.itnlim = 20
    t1 = time.perf_counter()
    r = s.solve_device(d_b.ptr.value, d_x.ptr.value, facts["damp"])
    t2 = time.perf_counter()
    tm = s.last_timing()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"pieces: set {1e6*(t1-t0):.1f}  solve {1e6*(t2-t1):.1f} (lib host clock {tm.solve_ms*1e3:.1f})  timing {1e6*(t3-t2):.1f}  sync {1e6*(t4-t3):.1f}", flush=True)
