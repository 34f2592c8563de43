#!/usr/bin/env python3
"""How far do the iteration counts of the reference's 18-problem suite move under rounding alone?

The suite runs LSQR to eps-level tolerances on operators with 25 distinct singular values repeated 40
times (test/lsqrtest_module.f90:55-94, 422-505): exact arithmetic would stop after 25 iterations and
rounding decides how many more it takes.  This script re-runs the pinned restatement
(oracle/lstp_oracle.c, the compiled reference's log digit for digit in the reference's own summation
order) with the dot product inside `hprod` summed in three other legal orders -- descending, pairwise,
eight interleaved partial sums -- and records every problem's iteration counts.  The GPU operator
(tree sums) is held to that spread, widened by its own width (tests/test_gpu_operator.py,
tests/test_fortran.py), like the EZ cases are held to the reference's drift under COO permutations.

    make -C oracle && python tests/golden/gen_lstp_band.py      # writes tests/golden/lstp_itn_band.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

ORDERS = {(0, 0): "ascending (the reference)", (1, 0): "descending", (2, 0): "pairwise tree",
          (3, 0): "eight interleaved partial sums", (2, 1): "pairwise tree, norms pairwise too",
          (3, 1): "eight interleaved partial sums, norms pairwise"}


def main():
    po = oracle.port()
    po.L.oracle_lstp_set_sum_order.argtypes = [__import__("ctypes").c_int]
    po.L.oracle_set_norm_order.argtypes = [__import__("ctypes").c_int]
    out = []
    for (m, n, nd, p, damp) in oracle.SUITE:
        rec = dict(m=m, n=n, nduplc=nd, npower=p, damp=damp, itn={}, istop={}, enorm={})
        for (o, no), name in ORDERS.items():
            po.L.oracle_lstp_set_sum_order(o)
            po.L.oracle_set_norm_order(no)
            r = po.lstp_test(m, n, nd, p, damp)
            rec["itn"][name] = int(r["itn"])
            rec["istop"][name] = int(r["istop"])
            rec["enorm"][name] = float(r["enorm"])
        po.L.oracle_lstp_set_sum_order(0)
        po.L.oracle_set_norm_order(0)
        out.append(rec)
        print(m, n, p, rec["itn"])
    with open(os.path.join(HERE, "lstp_itn_band.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
