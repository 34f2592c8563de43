#!/usr/bin/env python3
"""A 20-iteration solve at config 2: start kernels inside the first graph against plain launches ahead of it
(option start_eager), with and without the HIP events around the loop (option loop_events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()          # before liblsqrhip.so touches the device (as bench.py does)
import bench
from lsqr_amd import capi
K = int(os.environ.get("K", "20"))
s, d_b, facts, host = bench.build_workload(bench.HEADLINE, None, itnlim=K)
d_x = capi.DeviceBuffer(8 * facts["n"])
s.atol = s.btol = s.conlim = 0.0
s.set_option("graph_iters", min(K + (K & 1), 50))
def run(label):
    for _ in range(5):
        bench.timed_solve(s, d_b, d_x, facts["damp"], K)
    ts = sorted(bench.timed_solve(s, d_b, d_x, facts["damp"], K)[0] for _ in range(40))
    print(f"{label:40s} median {1e6*ts[len(ts)//2]:7.1f} us  min {1e6*ts[0]:7.1f} us  -> {K/ts[len(ts)//2]:8.0f} it/s", flush=True)
for rep in range(2):
    for se in (0, 1):
        for le in (1, 0):
            s.set_option("start_eager", se); s.set_option("loop_events", le)
            run(f"start_eager={se} loop_events={le}")
