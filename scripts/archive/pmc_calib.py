#!/usr/bin/env python3
"""Known byte counts for calibrating FETCH_SIZE / WRITE_SIZE per access width (run under rocprofv3 --pmc):
k_scale reads and writes n doubles with 8-byte-per-lane accesses; k_dot reads 2 n doubles with 16-byte ones;
k_copy: 8-byte loads and stores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lsqr_amd import capi, problems as P
from lsqr_amd.capi import check, lib
from lsqr_amd.solver import lsqr_solver_ez
import ctypes as C
p = P.poisson2d(8, 8)
s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
n = 100_000_000
x = capi.DeviceBuffer(8 * n)
y = capi.DeviceBuffer(8 * n)
check(lib().lsqrhip_dscal(s._h, n, 0.0, x.ptr))
check(lib().lsqrhip_dscal(s._h, n, 0.0, y.ptr))
r = C.c_double()
for _ in range(3):
    check(lib().lsqrhip_dscal(s._h, n, 1.0, x.ptr))          # k_scale: 8 B/lane load + store
    check(lib().lsqrhip_dcopy(s._h, n, x.ptr, y.ptr))        # k_copy: 8 B/lane
    check(lib().lsqrhip_ddot(s._h, n, x.ptr, y.ptr, C.byref(r)))   # k_dot: 16 B/lane loads
print("n =", n, "bytes per vector =", 8 * n)
