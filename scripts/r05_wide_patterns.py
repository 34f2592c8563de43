#!/usr/bin/env python3
"""Wide row patterns (pat.h, two-byte pattern numbers, the table through L2) against the layouts such a matrix had
before: a piecewise-constant-coefficient five-point mesh (tests/test_gpu_patterns.py piecewise_mesh), built once per
layout in ONE process, the products timed alternately.  usage: r05_wide_patterns.py [nx ny bx by] [reps] [rounds]"""
import os, sys, statistics, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from lsqr_amd.solver import lsqr_solver_ez
from test_gpu_patterns import piecewise_mesh

nx, ny, bx, by = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (4000, 4000, 16, 16)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 5
t0 = time.time()
m, n, irow, icol, a, b = piecewise_mesh(nx, ny, bx, by)
print(f"mesh {nx} x {ny}, {bx} x {by} regions: {m} rows, {a.size} nonzeros ({time.time() - t0:.1f} s on the host)", flush=True)
LAYOUTS = [("wide row patterns", {}),
           ("wide, slice form", {"LSQRHIP_PAT_PAIR": "0"}),
           ("wide, 6 workgroups per CU", {"LSQRHIP_PAT_GRID": "1536"}),
           ("without the wide table", {"LSQRHIP_PAT2": "0"}),
           ("no pattern layout", {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0"}),
           ("sliced ELL, 8-byte values", {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_SELLP": "0", "LSQRHIP_VAL8": "0"})]
solvers = []
for name, env in LAYOUTS:
    os.environ.update(env)
    t0 = time.time()
    s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=10)
    info = s.info()
    print(f"  {name:28s} sell {info['sell']} wide {info['pat_wide']:5d} value bytes {info['value_bytes']} "
          f"layout {info['csr_bytes'] / m:6.2f} bytes/row   create {time.time() - t0:.2f} s", flush=True)
    solvers.append((name, s, info))
    for k in env:
        os.environ.pop(k)
# every layout gives the same bits
x0 = np.cos(np.arange(n) * 0.37)
solvers_ok = solvers
ys = []
for name, s, info in solvers_ok:
    x, y = x0.copy(), np.zeros(m)
    s.aprod(1, m, n, x, y)
    ys.append(y)
print("  products bit-identical across the layouts:", all(np.array_equal(ys[0], y) for y in ys[1:]))
t = {name: ([], []) for name, _, _ in solvers}
for _ in range(rounds):
    for name, s, info in solvers:
        t[name][0].append(s.bench_kernel(1, reps))
        t[name][1].append(s.bench_kernel(2, reps))
for name, s, info in solvers:
    a1, a2 = t[name]
    lay1 = info["csr_bytes"] + 8 * n + 16 * m
    print(f"  {name:28s} mode 1: min {min(a1) * 1e3:7.1f} median {statistics.median(a1) * 1e3:7.1f} us "
          f"({lay1 / (min(a1) * 1e-3) / 1e9 / 8000:.3f} of 8 TB/s on {lay1 / 1e6:.0f} MB) | mode 2: min {min(a2) * 1e3:7.1f} us", flush=True)
