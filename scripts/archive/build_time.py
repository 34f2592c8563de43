#!/usr/bin/env python3
"""Time of initialize (COO already in HBM -> ready-to-solve handle) for a few workloads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen
for spec in sys.argv[1:] or ["poisson2d:1000:1000", "random:4000000:1000000:100", "powerlaw:5000000:2000000:10000"]:
    best = 1e9
    for k in range(3):
        dp = devgen.generate(spec)
        best = min(best, dp.solver.build_seconds)
        nnz = dp.nnz
        del dp
    print(f"{spec:36s} nnz {nnz:>11d}  initialize {best*1e3:9.2f} ms  ({nnz/best/1e9:6.2f} G nnz/s)", flush=True)
