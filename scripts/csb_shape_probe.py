#!/usr/bin/env python3
"""Layout facts and product times of column-swept workloads: row blocks, column splits, launches, mode 1 / mode 2 / update
(loop form), GB/s of layout bytes.  usage: csb_shape_probe.py SPEC..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen
for spec in sys.argv[1:]:
    dp = devgen.generate(spec)
    s = dp.solver
    info = s.info()
    g = lambda k: s.get_option(k)
    t = [min(s.bench_kernel(w, 20), s.bench_kernel(w, 20)) for w in (1, 2, 3)]
    lay1 = info["csr_bytes"] + 8 * dp.n + 16 * dp.nrows
    lay2 = info["csrt_bytes"] + 8 * dp.nrows + 16 * dp.n
    env = " ".join(f"{k[8:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("LSQRHIP_"))
    print(f"{spec:34s} [{env}] A: blocks {g('csb_blocks_mode1')} splits {g('csb_splits_mode1')} launches {g('launches_mode1')}"
          f" | A': blocks {g('csb_blocks_mode2')} splits {g('csb_splits_mode2')} launches {g('launches_mode2')}"
          f" | mode 1 {t[0]*1e3:8.1f} us {lay1/t[0]/1e6:6.0f} GB/s | mode 2 {t[1]*1e3:8.1f} us {lay2/t[1]/1e6:6.0f} GB/s | update {t[2]*1e3:6.1f} us",
          flush=True)
    del dp, s
