#!/usr/bin/env python3
"""One case of tests/fuzz_layouts.py in detail.  usage: fuzz_case.py SEED CASE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_layouts", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fuzz_layouts.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from lsqr_amd.solver import lsqr_solver_ez
seed, want = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(seed)
for case in range(want + 1):
    m, n, irow, icol, a, b = fz.make_case(rs)
    xp, yp = rs.uniform(-1, 1, size=n), rs.uniform(-1, 1, size=m)
po = oracle.port()
print("case", want, "m", m, "n", n, "nnz", irow.size, "lib", os.environ.get("LSQRHIP_LIB"))
for itn in (1, 2, 3, 4, 5, 6):
    o = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=itn)
    line = f"itnlim {itn}: oracle istop {o.istop} itn {o.itn} x {o.x[:2]} anorm {o.anorm:.6e} rnorm {o.rnorm:.6e} arnorm {o.arnorm:.3e}"
    for lay in ({}, {"LSQRHIP_CSB": "1"}):
        for k in fz.KNOBS:
            os.environ.pop(k, None)
        os.environ.update(lay)
        s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=itn)
        r = s.solve(b, 1e-2)
        line += f"\n      {str(lay):22s} istop {r.istop} itn {r.itn} x {r.x[:2]} anorm {r.anorm:.6e} rnorm {r.rnorm:.6e} arnorm {r.arnorm:.3e}"
    print(line, flush=True)
