#!/usr/bin/env python3
"""A/B of one LSQRHIP_* knob on ONE box, in ONE process: the workload is built once per value of the knob (knobs
are read at create), then the products of the variants are timed alternately, `rounds` times, and the minimum and
the median of each are printed.  (Boxes of the pool differ by +-10 % at HBM-resident sizes, and so do two
processes on one box a minute apart: variants measured in separate runs cannot be compared.)
usage: ab_env.py SPEC VAR=v1,v2[,v3] [reps=10] [rounds=5]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqr_amd import devgen

spec = sys.argv[1]
var, vals = sys.argv[2].split("=")
vals = vals.split(",")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
solvers = []
for v in vals:
    if v == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = v
    dp = devgen.generate(spec)
    solvers.append((v, dp))
os.environ.pop(var, None)
t = {v: ([], []) for v in vals}
for _ in range(rounds):
    for v, dp in solvers:
        t[v][0].append(dp.solver.bench_kernel(1, reps))
        t[v][1].append(dp.solver.bench_kernel(2, reps))
for v in vals:
    a, b = t[v]
    print(f"{spec:32s} {var}={v:4s} mode 1: min {min(a)*1e3:8.1f} median {statistics.median(a)*1e3:8.1f} us | "
          f"mode 2: min {min(b)*1e3:8.1f} median {statistics.median(b)*1e3:8.1f} us", flush=True)
