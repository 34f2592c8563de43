#!/usr/bin/env python3
"""Is the mode-1 / mode-2 difference of a square random system an artefact of measurement order?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen
spec = sys.argv[1]
dp = devgen.generate(spec)
s = dp.solver
for order in ((2, 1), (1, 2), (2, 1)):
    t = {w: s.bench_kernel(w, 10) for w in order}
    print(spec, "order", order, {w: round(t[w] * 1e3, 1) for w in (1, 2)}, flush=True)
