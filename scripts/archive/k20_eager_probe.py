#!/usr/bin/env python3
"""A 20-iteration solve at config 2: one graph launch against eager launches, and graph batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lsqr_amd import capi
import torch
s, d_b, facts, host = bench.build_workload(bench.HEADLINE, None, itnlim=20)
d_x = capi.DeviceBuffer(8 * facts["n"])
s.atol = s.btol = s.conlim = 0.0
def run(label):
    for _ in range(5):
        bench.timed_solve(s, d_b, d_x, facts["damp"], 20)
    ts = []
    for _ in range(30):
        dt, r, restarts = bench.timed_solve(s, d_b, d_x, facts["damp"], 20)
        ts.append(dt)
    ts.sort()
    print(f"{label:28s} median {1e6*ts[len(ts)//2]:7.1f} us  min {1e6*ts[0]:7.1f} us  -> {20/ts[len(ts)//2]:8.0f} it/s", flush=True)
for gi in (20, 10, 6, 4, 2):
    s.set_option("graph", 1); s.set_option("graph_iters", gi)
    run(f"graph, batches of {gi}")
s.set_option("graph", 0)
run("eager launches")
s.set_option("poll_ahead", 0)
run("eager, no look-ahead")
