#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference.

Run in the build container only (needs /root/reference to build oracle/_ref):

    make -C oracle && python tests/golden/gen_golden.py

Every expected value below is produced by the unmodified reference
(/root/reference/src compiled by oracle/Makefile, amdflang -O2 -ffp-contract=off)
through oracle/ref_shim.f90.  Inputs are regenerated from seeds by
lsqr_amd.problems (their sha256 prefix is stored to catch generator drift);
floats are stored as C99 hex strings so the fixtures are bit-exact.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from cases import build_cases  # noqa: E402
from lsqr_amd import problems as P  # noqa: E402


def hx(v):
    if isinstance(v, np.ndarray):
        return [float(t).hex() for t in v]
    return float(v).hex()


def blas_vectors():
    """Named input vectors for the BLAS-1 goldens (src/lsqrblas.f90)."""
    u = lambda seed, n: P.u64_to_unit(P.rng_u64(seed, 9, np.arange(n, dtype=np.uint64)))
    v = {
        "empty": np.zeros(0),
        "single_neg": np.array([-3.5]),
        "zeros": np.zeros(17),
        "small7": u(1, 7),
        "n5": u(2, 5),
        "n4": u(3, 4),
        "n1001": u(4, 1001),
        "huge": u(5, 64) * 1e200,        # x**2 overflows: exercises the scaling in dnrm2
        "tiny": u(6, 64) * 1e-200,       # x**2 underflows
        "mixed": np.concatenate([u(7, 30) * 1e150, u(8, 30) * 1e-150, np.zeros(4)]),
    }
    return v


def main():
    rf = oracle.ref()
    if rf is None:
        raise SystemExit("oracle/_ref/libref_lsqr.so missing: run `make -C oracle` where /root/reference exists")

    # ---- solve goldens ------------------------------------------------------
    out = {}
    for name, (p, o) in build_cases().items():
        r = rf.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
        rec = dict(problem=p.name, m=p.m, n=p.n, nnz=p.nnz, checksum=P.checksum(p), options=o,
                   istop=r.istop, itn=r.itn, x=hx(r.x), anorm=hx(r.anorm), acond=hx(r.acond),
                   arnorm=hx(r.arnorm), xnorm=hx(r.xnorm))
        # rnorm is never assigned by the reference when the loop is skipped
        # (src/lsqr.f90:646-653): do not pin garbage.
        rec["rnorm"] = hx(r.rnorm) if r.istop != 0 else None
        if o["wantse"]:
            rec["se"] = hx(r.se)
        # The reference's OWN sensitivity to summation order: rerun it on the same
        # matrix with the COO triplets permuted (legal input, src/lsqr.f90:168-172 sums
        # in COO order).  Long / ill-conditioned runs drift far above 1e-10 by
        # themselves; parity tests use tol = max(1e-10, 10 * this band).
        if p.nnz > 1:
            sx = sa = sr = 0.0
            itns, istops = {r.itn}, {r.istop}
            for seed in range(1, 13):   # 12 permutations: itn at a tolerance crossing needs a real sample
                q = P.shuffled(p, seed)
                rq = rf.solve(q.m, q.n, q.irow, q.icol, q.a, q.b, **o)
                nx = float(np.linalg.norm(r.x))
                sx = max(sx, float(np.linalg.norm(rq.x - r.x)) / nx if nx > 0 else 0.0)
                sa = max(sa, abs(rq.anorm - r.anorm) / r.anorm if r.anorm > 0 else 0.0)
                sr = max(sr, abs(rq.rnorm - r.rnorm) / r.rnorm if r.rnorm > 0 and r.istop != 0 else 0.0)
                itns.add(rq.itn)
                istops.add(rq.istop)
            rec["sens"] = dict(x=sx, anorm=sa, rnorm=sr, itn=sorted(itns), istop=sorted(istops))
        # aprod mode 1 / 2 on this matrix with fixed probe vectors (src/lsqr.f90:134-200)
        if p.nnz > 0:
            xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
            yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
            _, y1 = rf.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
            x2, _ = rf.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
            rec["aprod1_y"] = hx(y1)
            rec["aprod2_x"] = hx(x2)
            rec["acheck_inform"] = rf.acheck(p.m, p.n, p.irow, p.icol, p.a)
            inform, tests, u, v, w = rf.xcheck(p.m, p.n, p.irow, p.icol, p.a, r.anorm, o["damp"], p.b, r.x)
            rec["xcheck"] = dict(inform=inform, tests=hx(tests), u_norm=hx(rf.dnrm2(u)) if p.m else None,
                                 w_norm=hx(rf.dnrm2(w)) if p.n else None)
        out[name] = rec
        print(f"{name:26s} istop={r.istop} itn={r.itn}")
    with open(os.path.join(HERE, "solve_cases.json"), "w") as f:
        f.write("{\n" + ",\n".join(json.dumps(k) + ":" + json.dumps(v, separators=(",", ":"))
                                   for k, v in out.items()) + "\n}\n")

    # ---- BLAS-1 goldens ------------------------------------------------------
    vecs = blas_vectors()
    bl = {}
    for k, x in vecs.items():
        rec = dict(n=len(x), dnrm2=hx(rf.dnrm2(x, 1, len(x))) if True else None)
        if len(x) >= 4:
            rec["dnrm2_inc2"] = hx(rf.dnrm2(x, 2, (len(x) + 1) // 2))
        y = x[::-1].copy()
        rec["ddot_rev"] = hx(rf.ddot(x, y)) if len(x) else hx(0.0)
        if np.all(np.isfinite(x * -0.37)):
            rec["dscal_m037"] = hx(rf.dscal(-0.37, x))
        rec["dcopy_ok"] = bool(np.array_equal(rf.dcopy(x), x)) if len(x) else True
        bl[k] = rec
    with open(os.path.join(HERE, "blas1.json"), "w") as f:
        f.write("{\n" + ",\n".join(json.dumps(k) + ":" + json.dumps(v, separators=(",", ":"))
                                   for k, v in bl.items()) + "\n}\n")

    # ---- iteration-log golden (nout /= 0; src/lsqr.f90:589-595, 655-671, 813-837, 872-880) ----
    from cases import TRUNCATED_TWINS
    for name in ["t1_readme_default", "random_over_damped"] + TRUNCATED_TWINS:
        p, o = build_cases()[name]
        with tempfile.TemporaryDirectory() as td:
            lp = os.path.join(td, "log.txt")
            rf.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, logpath=lp, **o)
            txt = open(lp).read()
        with open(os.path.join(HERE, f"log_{name}.txt"), "w") as f:
            f.write(txt)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
