#!/usr/bin/env python3
"""Wall time of lsqrhip_create_sharded (what Fortran's initialize(..., ngpu=N) calls) on a host COO system, by number of
row blocks, in the loopback harness (all ranks on this GPU).  usage: sharded_init_time.py SPEC [ngpu ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LSQRHIP_SHARD_LOOPBACK"] = "1"
import numpy as np
from lsqr_amd import devgen, capi
from lsqr_amd.capi import check, lib
spec = sys.argv[1]
ngpus = [int(t) for t in sys.argv[2:]] or [1, 2, 4, 8]
cfg = devgen.parse_spec(spec)
irow, icol, a, b = devgen.download_coo(spec)
print(f"{spec}: nnz {len(a)}  lib {os.environ.get('LSQRHIP_LIB', 'liblsqrhip.so')}", flush=True)
for P in ngpus:
    h = C.c_void_p()
    t0 = time.perf_counter()
    check(lib().lsqrhip_create_sharded(cfg["m"], cfg["n"], a.size, irow.ctypes.data, icol.ctypes.data, a.ctypes.data, P, C.byref(h)))
    dt = time.perf_counter() - t0
    check(lib().lsqrhip_destroy(h))
    print(f"  ngpu {P}: create_sharded {dt:7.2f} s", flush=True)
