#!/usr/bin/env python3
"""13 mode-1 products, then 13 mode-2 products of one generated workload (3 warm + 10 each): the program
scripts/pmc_csb.sh runs under rocprofv3 --pmc.  usage: pmc_products.py SPEC"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen
spec = sys.argv[1]
dp = devgen.generate(spec)
s = dp.solver
g = s.get_option
# ("splits" as the summary script uses them: 1 where the splits are closed inside the sweep launch -- no combine dispatch to
#  subtract; the real numbers follow as "S")
f1, f2 = g("csb_fuse_mode1"), g("csb_fuse_mode2")
print("LAYOUT", spec, "launches", g("dispatches_mode1"), g("dispatches_mode2"), "blocks", g("csb_blocks_mode1"),
      g("csb_blocks_mode2"), "splits", 1 if f1 else g("csb_splits_mode1"), 1 if f2 else g("csb_splits_mode2"), "bytes", s.info()["csr_bytes"],
      s.info()["csrt_bytes"], "S", g("csb_splits_mode1"), g("csb_splits_mode2"), "fused", f1, f2,
      "xfold" if g("launches_mode1") == g("dispatches_mode1") else "xpass", flush=True)
t1 = s.bench_kernel(1, 10)
t2 = s.bench_kernel(2, 10)
print("TIMES_MS", round(t1, 4), round(t2, 4), flush=True)
