#!/usr/bin/env python3
"""Config 2, K = 2000: iterations per captured graph batch (option graph_iters) against iterations/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench
from lsqr_amd import capi
K = 2000
s, d_b, facts, host = bench.build_workload(bench.HEADLINE, None, itnlim=K)
d_x = capi.DeviceBuffer(8 * facts["n"])
s.atol = s.btol = s.conlim = 0.0
for gi in (50, 100, 200, 250, 500, 1000, 100):
    s.set_option("graph_iters", gi)
    for _ in range(2):
        bench.timed_solve(s, d_b, d_x, facts["damp"], K)
    ts = sorted(bench.timed_solve(s, d_b, d_x, facts["damp"], K)[0] for _ in range(5))
    print(f"graph_iters {gi:5d}: median {1e3*ts[2]:8.3f} ms -> {K/ts[2]:8.0f} it/s", flush=True)
