import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from test_gpu_patterns import stencil, _vec
from lsqr_amd.solver import lsqr_solver_ez
m, n, irow, icol, a, b = stencil(50000, 50000, (-3, -1, 0, 1, 3), (1.0,) * 5)
a = np.choose((_vec(11, a.size) * 3).astype(int).clip(0, 2), [2.0, -1.0, 0.5])
for k in range(5):
    s = lsqr_solver_ez().initialize(m, n, a, irow, icol)
    print(k, s.info()["sell"], flush=True)
