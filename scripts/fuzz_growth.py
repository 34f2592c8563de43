#!/usr/bin/env python3
"""How the distance between the GPU's x and the oracle's grows iteration by iteration on one fuzz case, next to the
oracle's own distance under a permutation of its input.  usage: fuzz_growth.py SEED CASE"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import fuzz_layouts as fz
from lsqr_amd.solver import lsqr_solver_ez
seed, want = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(seed)
for case in range(want + 1):
    m, n, irow, icol, a, b = fz.make_case(rs)
    xp, yp = rs.uniform(-1, 1, size=n), rs.uniform(-1, 1, size=m)
po = oracle.port()
print("case", want, "m", m, "n", n, "nnz", irow.size)
for lay in ({}, {"LSQRHIP_CSB": "1"}, {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_SELL": "0"}):
    for k in fz.KNOBS:
        os.environ.pop(k, None)
    os.environ.update(lay)
    for itn in range(1, 9):
        o = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=itn)
        ds = []
        for k in range(6):
            perm = np.random.RandomState(5000 + k).permutation(irow.size)
            o2 = po.solve(m, n, irow[perm], icol[perm], a[perm], b, damp=1e-2, itnlim=itn)
            ds.append(np.linalg.norm(o2.x - o.x) / np.linalg.norm(o.x))
        s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=itn)
        r = s.solve(b, 1e-2)
        e = np.linalg.norm(r.x - o.x) / np.linalg.norm(o.x)
        print(f"{lay} itn {itn}: gpu vs oracle {e:.2e}  oracle under 6 permutations {min(ds):.2e} .. {max(ds):.2e}  "
              f"anorm {r.anorm:.15e}/{o.anorm:.15e} rnorm {r.rnorm:.12e}/{o.rnorm:.12e}", flush=True)
