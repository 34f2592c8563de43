#!/usr/bin/env python3
"""A/B of one LSQRHIP_* knob on whole SOLVES, on one box in one process (see ab_env.py for why): the workload is built
once per value of the knob, then K-iteration solves of the variants are timed alternately, `rounds` times; prints the
minimum and the median ms per iteration of each, and whether the variants returned the same x bit for bit.
usage: ab_solve.py SPEC VAR=v1,v2[,v3] [K=200] [rounds=5]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lsqr_amd import devgen
from lsqr_amd.capi import DeviceBuffer

spec = sys.argv[1]
var, vals = sys.argv[2].split("=")
vals = vals.split(",")
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
probs = []
for v in vals:
    if v == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = v
    dp = devgen.generate(spec)
    dp.solver.atol = dp.solver.btol = dp.solver.conlim = 0.0
    dp.solver.itnlim = K
    probs.append((v, dp, DeviceBuffer(8 * dp.n)))
os.environ.pop(var, None)
t = {v: [] for v in vals}
xs = {}
for rnd in range(rounds + 1):
    for v, dp, d_x in probs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = dp.solver.solve_device(dp.d_b.ptr.value, d_x.ptr.value, dp.damp)
        torch.cuda.synchronize()
        if rnd:
            t[v].append((time.perf_counter() - t0) / max(r.itn, 1) * 1e3)
        else:
            xs[v] = d_x.to_array(np.float64, dp.n).copy()
for v in vals:
    same = all(np.array_equal(xs[v].view(np.uint64), xs[w].view(np.uint64)) for w in vals)
    print(f"{spec:32s} {var}={v:4s} K={K}: min {min(t[v]):8.4f} median {statistics.median(t[v]):8.4f} ms/iteration"
          f"   x identical across variants: {same}", flush=True)
