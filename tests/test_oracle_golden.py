"""Pin the CPU checker (oracle/lsqr_oracle.c) against the reference.

* bit-for-bit against tests/golden/*.json (produced by the compiled reference,
  tests/golden/gen_golden.py) -- runs anywhere;
* bit-for-bit against the live reference library oracle/_ref when it is present.

CPU only: these run under `-m "not gpu"`.
"""
import json
import os

import numpy as np
import pytest

import oracle
from cases import build_cases
from golden.gen_golden import blas_vectors
from lsqr_amd import problems as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SOLVE = json.load(open(os.path.join(GOLD, "solve_cases.json")))
BLAS = json.load(open(os.path.join(GOLD, "blas1.json")))
CASES = build_cases()


def fh(s):
    return float.fromhex(s)


def fhv(lst):
    return np.array([float.fromhex(t) for t in lst], dtype=np.float64)


def test_case_lists_match():
    assert set(SOLVE) == set(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_generators_reproduce_fixture_inputs(name):
    p, o = CASES[name]
    g = SOLVE[name]
    assert (g["m"], g["n"], g["nnz"]) == (p.m, p.n, p.nnz)
    assert g["checksum"] == P.checksum(p), "lsqr_amd.problems drifted from the committed fixtures"
    assert g["options"] == o


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_solve_bit_exact_vs_golden(name):
    p, o = CASES[name]
    g = SOLVE[name]
    r = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
    assert r.istop == g["istop"]
    assert r.itn == g["itn"]
    assert np.array_equal(r.x, fhv(g["x"]))
    for k in ("anorm", "acond", "arnorm", "xnorm"):
        assert getattr(r, k) == fh(g[k]), k
    if g["rnorm"] is not None:
        assert r.rnorm == fh(g["rnorm"])
    else:
        # reference leaves rnorm unassigned when no iteration runs; the checker
        # (and the HIP path) define it as beta = norm(b)  [SURVEY.md 8b quirk]
        assert r.rnorm == oracle.port().dnrm2(p.b) if p.m > 1 else True
    if o["wantse"]:
        assert np.array_equal(r.se, fhv(g["se"]))


@pytest.mark.parametrize("name", sorted(k for k, v in CASES.items() if v[0].nnz > 0))
def test_oracle_aprod_acheck_xcheck_vs_golden(name):
    p, o = CASES[name]
    g = SOLVE[name]
    po = oracle.port()
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    x1, y1 = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.array_equal(x1, xp), "mode 1 must not change x"
    assert np.array_equal(y1, fhv(g["aprod1_y"]))
    x2, y2 = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.array_equal(y2, yp), "mode 2 must not change y"
    assert np.array_equal(x2, fhv(g["aprod2_x"]))
    inform, err = po.acheck(p.m, p.n, p.irow, p.icol, p.a)
    assert inform == g["acheck_inform"] == 0
    assert err <= 1e-14
    xs = fhv(g["x"])
    inform, tests, u, v, w = po.xcheck(p.m, p.n, p.irow, p.icol, p.a, fh(g["anorm"]), o["damp"], p.b, xs)
    assert inform == g["xcheck"]["inform"]
    assert np.array_equal(tests, fhv(g["xcheck"]["tests"]))
    assert po.dnrm2(u) == fh(g["xcheck"]["u_norm"])
    assert po.dnrm2(w) == fh(g["xcheck"]["w_norm"])


def test_aprod_invalid_mode():
    p = P.readme_3x3()
    with pytest.raises(ValueError):
        oracle.port().aprod(3, p.m, p.n, p.irow, p.icol, p.a, np.zeros(3), np.zeros(3))


def test_validate_matches_reference_checks():
    # src/lsqr.f90:110-111: only upper bounds are checked
    po = oracle.port()
    assert po.validate(3, 3, [1, 2, 3], [1, 2, 3]) == 0
    assert po.validate(3, 3, [1, 4, 3], [1, 2, 3]) == 2
    assert po.validate(3, 3, [1, 2, 3], [1, 2, 9]) == 3
    assert po.validate(3, 3, [0, 2, 3], [1, 2, 3]) == 0  # lower bound not checked by the reference


@pytest.mark.parametrize("name", sorted(BLAS))
def test_oracle_blas1_bit_exact_vs_golden(name):
    po = oracle.port()
    x = blas_vectors()[name]
    g = BLAS[name]
    assert len(x) == g["n"]
    assert po.dnrm2(x, 1, len(x)) == fh(g["dnrm2"])
    if "dnrm2_inc2" in g:
        assert po.dnrm2(x, 2, (len(x) + 1) // 2) == fh(g["dnrm2_inc2"])
    d, gd = po.ddot(x, x[::-1].copy()), fh(g["ddot_rev"])
    assert d == gd or (np.isnan(d) and np.isnan(gd))   # 'huge': inf - inf in both
    if "dscal_m037" in g:
        assert np.array_equal(po.dscal(-0.37, x), fhv(g["dscal_m037"]))
    assert np.array_equal(po.dcopy(x), x)


def test_d2norm_properties():
    # src/lsqr.f90:1164-1179 (private in the reference: pinned through anorm/rnorm above)
    po = oracle.port()
    assert po.d2norm(0.0, 0.0) == 0.0
    assert abs(po.d2norm(3.0, 4.0) - 5.0) <= 4 * np.finfo(float).eps * 5.0   # scaled form is not exact
    assert po.d2norm(-3.0, 4.0) == po.d2norm(3.0, 4.0)
    assert np.isfinite(po.d2norm(1e300, 1e300))
    assert po.d2norm(1e-300, 1e-300) > 0.0


def test_readme_published_result():
    """README.md:55-58: istop = 1, x = 1.242424E+00 -6.060606E-02 -4.040404E-02;
    test/lsqrtest_ez.f90:50: |A x - b| <= 1e-12."""
    p = P.readme_3x3()
    r = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b)
    assert r.istop == 1
    assert np.allclose(r.x, [1.242424, -6.060606e-2, -4.040404e-2], rtol=0, atol=5e-7)
    assert np.max(np.abs(p.dense() @ r.x - p.b)) <= 1e-12
    p2 = P.ez_3x4()
    r2 = oracle.port().solve(p2.m, p2.n, p2.irow, p2.icol, p2.a, p2.b)
    assert r2.istop == 1
    assert np.max(np.abs(p2.dense() @ r2.x - p2.b)) <= 1e-12   # test/lsqrtest_ez.f90:102


def test_log_records_match_reference_log_text():
    """The per-iteration record the checker emits equals what the reference prints
    (format '(1P, I6, 2E17.9, 4E10.2, E9.1, 3E8.1)', src/lsqr.f90:828-829)."""
    p, o = CASES["t1_readme_default"]
    r = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, want_log=True, **o)
    lines = [l for l in open(os.path.join(GOLD, "log_t1_readme_default.txt")).read().splitlines()
             if l[:6].strip().isdigit() and int(l[:6]) >= 1]
    assert len(lines) == r.itn == len(r.log)
    for line, rec in zip(lines, r.log):
        itn = int(line[:6])
        x1 = float(line[6:23])
        rn = float(line[23:40])
        assert itn == int(rec[0])
        assert abs(x1 - rec[1]) <= 5e-10 * max(1.0, abs(rec[1]))
        assert abs(rn - rec[2]) <= 5e-10 * max(abs(rec[2]), 1e-300) * 1.0001 + 1e-300


@pytest.mark.skipif(oracle.ref() is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_bit_exact_vs_live_reference(name):
    p, o = CASES[name]
    a = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
    b = oracle.ref().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
    assert (a.istop, a.itn) == (b.istop, b.itn)
    assert np.array_equal(a.x, b.x)
    assert (a.anorm, a.acond, a.arnorm, a.xnorm) == (b.anorm, b.acond, b.arnorm, b.xnorm)
    if a.istop != 0:
        assert a.rnorm == b.rnorm
    if o["wantse"]:
        assert np.array_equal(a.se, b.se)
