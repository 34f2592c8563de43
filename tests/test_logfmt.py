"""Host logic: the nout /= 0 log formatter reproduces the reference's log text
(tests/golden/log_*.txt, written by the compiled reference) character for character
when fed the checker's per-iteration records.  CPU only."""
import os

import numpy as np
import pytest

import oracle
from cases import build_cases
from lsqr_amd.logfmt import format_log

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def extras_from_records(p, rec):
    po = oracle.port()
    bnorm = po.dnrm2(p.b)
    u = po.dscal(1.0 / bnorm, p.b)
    v, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, np.zeros(p.n), u)
    alpha0 = po.dnrm2(v)
    dxk = rec[:, 9]
    dxmax, maxdx = 0.0, 0
    for i, d in enumerate(dxk):      # src/lsqr.f90:754-757 (strict <)
        if dxmax < d:
            dxmax, maxdx = d, i + 1
    return dict(bnorm=bnorm, beta0=bnorm, test2_0=alpha0 / bnorm, dxmax=dxmax, maxdx=maxdx)


@pytest.mark.parametrize("name", ["t1_readme_default", "random_over_damped"])
def test_formatter_matches_reference_log(name):
    p, o = build_cases()[name]
    r = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, want_log=True, **o)
    text = format_log(p.m, p.n, o["damp"], o["wantse"], o["atol"], o["btol"], o["conlim"], o["itnlim"],
                      r.log, r, **extras_from_records(p, r.log))
    want = open(os.path.join(GOLD, f"log_{name}.txt")).read()
    assert text.splitlines() == want.splitlines()
