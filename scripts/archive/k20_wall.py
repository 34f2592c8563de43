#!/usr/bin/env python3
"""Wall time of a K-iteration solve at config 2 (K from the environment, default 20) under the LSQRHIP_* environment
in force: median and minimum of 60 solves."""
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
for _ in range(8):
    bench.timed_solve(s, d_b, d_x, facts["damp"], K)
raw = [bench.timed_solve(s, d_b, d_x, facts["damp"], K)[0] for _ in range(60)]
if os.environ.get("K20_RAW"):
    print("in order, us:", " ".join(f"{1e6*t:.0f}" for t in raw), flush=True)
ts = sorted(raw)
env = " ".join(f"{k[8:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("LSQRHIP_"))
print(f"K={K} [{env:24s}] median {1e6*ts[len(ts)//2]:7.1f} us  min {1e6*ts[0]:7.1f} us  -> {K/ts[len(ts)//2]:8.0f} it/s", flush=True)
