#!/usr/bin/env python3
"""Phase clocks of the column-swept product's launches (LSQRHIP_CSB_PROBE=1; csb.h CsbMat.probe): how long the
coefficients, the grids + clear, the sweep, the publication of the splits' sums and the epilogue take, per workgroup.
usage: csb_probe.py SPEC [ENV=VAL ...]     (measurement only)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LSQRHIP_CSB_PROBE"] = "1"
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    os.environ[k] = v
import numpy as np
from lsqr_amd import capi, devgen

spec = sys.argv[1]
dp = devgen.generate(spec)
s = dp.solver
L, W = 8, 512
for mode in (1, 2):
    ptr = s.get_option(f"csb_probe_mode{mode}")
    if not ptr:
        print(f"mode {mode}: not a column-swept layout")
        continue
    ms = s.bench_kernel(mode, 5)      # the LAST launch's clocks stay in the buffer
    buf = np.zeros(L * W * 8, dtype=np.uint64)
    capi.check(capi.lib().lsqrhip_dev_download(buf.ctypes.data, ptr, buf.nbytes))
    buf = buf.reshape(L, W, 8).astype(np.int64)
    print(f"{spec} {' '.join(sys.argv[2:])} mode {mode}: {ms*1e3:.1f} us per product; splits {s.get_option(f'csb_splits_mode{mode}')}, "
          f"blocks {s.get_option(f'csb_blocks_mode{mode}')}, launches {s.get_option(f'launches_mode{mode}')}, fuse {s.get_option(f'csb_fuse_mode{mode}')}")
    for l in range(L):
        b = buf[l]
        live = b[:, 0] > 0
        if not live.any():
            continue
        b = b[live]
        t0 = b[:, 0].min()
        tick = 0.01      # us per tick (100 MHz)
        def col(k, sel=None):
            v = b[:, k] if sel is None else b[sel, k]
            v = v[v > 0]
            return (v - t0) * tick
        def fmt(v):
            return "   -  " if len(v) == 0 else f"{np.median(v):7.1f} [{v.min():6.1f} {v.max():6.1f}]"
        last = b[:, 6] == 1
        print(f"  launch {l}: {live.sum()} workgroups | entry {fmt(col(0))} | coef {fmt(col(1))} | sweep begins {fmt(col(2))} | "
              f"sweep done {fmt(col(3))} | published {fmt(col(4))} | closed {fmt(col(5))}" +
              (f" | closers {last.sum()}: published {fmt(col(4, last))} closed {fmt(col(5, last))}" if last.any() else ""))
