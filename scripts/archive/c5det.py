import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from lsqr_amd import devgen
from lsqr_amd.capi import DeviceBuffer
spec = sys.argv[1] if len(sys.argv) > 1 else "powerlaw:5000000:2000000:10000"
dp = devgen.generate(spec, itnlim=12)
s = dp.solver
print(s.info())
for k in ("csb_blocks_mode1","csb_blocks_mode2","csb_splits_mode1","csb_splits_mode2"):
    print(k, s.get_option(k))
d_x = DeviceBuffer(8 * dp.n)
for pipeline in (2, 2, 1, 1, 0, 2):
    s.set_option("pipeline", pipeline)
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    x = d_x.to_array(np.float64, dp.n)
    print(pipeline, r.istop, r.itn, repr(r.anorm), repr(r.rnorm), repr(float(np.linalg.norm(x))))
