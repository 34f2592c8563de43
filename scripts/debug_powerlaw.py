import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from lsqr_amd import problems as P
from lsqr_amd.solver import lsqr_solver_ez

dmax = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
p = P.powerlaw_rows(20000, 8000, seed=21, dmin=4, dmax=dmax)
print("nnz", p.nnz, "max row", np.max(np.bincount(p.irow)), "max col", np.max(np.bincount(p.icol)))
po = oracle.port()
for graph in (1, 0):
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=6, nout=os.devnull)
    s.set_option("graph", graph)
    r = s.solve(p.b, 0.0)
    rec = s.log_records()
    o = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, itnlim=6, want_log=True)
    print("graph", graph, "istop", r.istop, o.istop)
    for a, b in zip(rec, o.log):
        print(" itn %d  rnorm %.15e / %.15e  anorm %.15e / %.15e  x1 %.12e / %.12e dk %.6e/%.6e" % (a[0], a[2], b[2], a[5], b[5], a[1], b[1], a[8], b[8]))
