#!/usr/bin/env python3
"""Print back-to-back average kernel times (us) and GB/s for one workload under the current
LSQRHIP_* tuning environment.  usage: kernel_times.py SPEC [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen

spec = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dp = devgen.generate(spec)
s = dp.solver
info = s.info()
P = info["rowptr_bytes"]
m, n, nnz = dp.nrows, dp.n, dp.nnz
b1 = 12 * nnz + P * (m + 1) + 8 * n + 16 * m
b2 = 12 * nnz + P * (n + 1) + 8 * m + 16 * n
b3 = 40 * n
t = [s.bench_kernel(w, reps) for w in (1, 2, 3)]
t = [min(a, s.bench_kernel(w, reps)) for a, w in zip(t, (1, 2, 3))]
env = " ".join(f"{k[8:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("LSQRHIP_"))
lay = f"A:sell{info['sell']}/xl{info['xlds']}/P{info['panels']} A':sell{info['sell_t']}/xl{info['xlds_t']}/P{info['panels_t']}"
print(f"{spec:34s} [{env:28s}] {lay} spmv1 {t[0]*1e3:8.2f} us {b1/t[0]/1e6:7.0f} GB/s | spmv2 {t[1]*1e3:8.2f} us {b2/t[1]/1e6:7.0f} GB/s"
      f" | update {t[2]*1e3:7.2f} us {b3/t[2]/1e6:7.0f} GB/s", flush=True)
