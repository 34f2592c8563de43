#!/usr/bin/env python3
"""mode 1 vs mode 2 of a square random system through lsqrhip_aprod_device on caller-owned vectors
(fresh allocations, roles swapped between runs): is the asymmetry in the vectors or in the streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lsqr_amd import capi, devgen
from lsqr_amd.capi import check, lib
spec = sys.argv[1]
dp = devgen.generate(spec)
s = dp.solver
n, m = dp.n, dp.m
bufs = [capi.DeviceBuffer.from_array(np.full(max(m, n), 1e-3)) for _ in range(4)]
def t(mode, bx, by, reps=6):
    check(lib().lsqrhip_aprod_device(s._h, mode, bx.ptr, by.ptr))
    check(lib().lsqrhip_dev_sync())
    t0 = time.perf_counter()
    for _ in range(reps):
        check(lib().lsqrhip_aprod_device(s._h, mode, bx.ptr, by.ptr))
    check(lib().lsqrhip_dev_sync())
    return (time.perf_counter() - t0) / reps * 1e3
for (ix, iy) in ((0, 1), (1, 0), (2, 3), (3, 2)):
    print(spec, f"x=buf{ix} y=buf{iy}: mode1 {t(1, bufs[ix], bufs[iy]):.3f} ms   mode2 {t(2, bufs[ix], bufs[iy]):.3f} ms", flush=True)
