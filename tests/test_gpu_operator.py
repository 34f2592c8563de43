"""LSQR on a device-resident user operator (csrc/op_api.h, lsqr_amd/operator.py) and the
reference's own test-problem class on the GPU.

The checker is oracle/lstp_oracle.c, itself pinned digit for digit against the compiled
reference's 18-problem log (tests/test_lstp_oracle.py)."""
import ctypes as C
import io
import json
import os

import numpy as np
import pytest

import oracle
from lsqr_amd import capi
from lsqr_amd import problems as P
from lsqr_amd.capi import check, lib
from lsqr_amd.operator import SUITE, lsqr_solver_device, run_suite, saunders_problem
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu


def test_suite_definition_is_the_references():
    assert SUITE == oracle.SUITE and len(SUITE) == 18


@pytest.mark.parametrize("case", [(2000, 1000, 40, 4, 1e-10), (1000, 1000, 40, 2, 1e-8), (1000, 2000, 40, 7, 1e-13),
                                  (37, 91, 5, 3, 1e-3), (64, 64, 1, 2, 0.0)])
def test_lstp_generator_and_operator_match_the_oracle(case):
    m, n, nd, p, damp = case
    g = oracle.port().lstp_generate(m, n, nd, p, damp)
    s = saunders_problem(m, n, nd, p, damp)
    for k in ("d", "hy", "hz", "xtrue", "b"):                 # same formulas, same order: same numbers
        np.testing.assert_allclose(getattr(s, k), g[k], rtol=0, atol=2e-16 * max(1.0, np.max(np.abs(g[k]))))
    assert s.acond_lstp == pytest.approx(g["acond"], rel=1e-15) and s.rnorm_lstp == pytest.approx(g["rnorm"], rel=1e-14)
    # aprod1 / aprod2 on the device (tree sums in hprod) against the sequential restatement
    xp = P.u64_to_unit(P.rng_u64(11, 1, np.arange(n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(12, 1, np.arange(m, dtype=np.uint64)))
    po = oracle.port()
    ctx = oracle_ctx(po, g, m, n)
    for mode in (1, 2):
        x, y = xp.copy(), yp.copy()
        s.aprod(mode, m, n, x, y)
        xo, yo = xp.copy(), yp.copy()
        po.L.oracle_lstp_aprod(C.byref(ctx), mode, m, n, xo.ctypes.data, yo.ctypes.data)
        scale = max(np.max(np.abs(xo)), np.max(np.abs(yo)), 1.0)
        assert np.max(np.abs(x - xo)) <= 1e-14 * scale and np.max(np.abs(y - yo)) <= 1e-14 * scale
    inform, err = s.acheck()
    assert inform == 0 and err < 1e-14


class _LstpCtx(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("d", C.c_void_p), ("hy", C.c_void_p), ("hz", C.c_void_p),
                ("w", C.c_void_p)]


def oracle_ctx(po, g, m, n):
    """oracle_lstp_t over the generated arrays (kept alive on the struct)."""
    ctx = _LstpCtx()
    ctx.m, ctx.n = m, n
    ctx._keep = (np.ascontiguousarray(g["d"]), np.ascontiguousarray(g["hy"]), np.ascontiguousarray(g["hz"]),
                 np.zeros(max(m, n)))
    ctx.d, ctx.hy, ctx.hz, ctx.w = (a.ctypes.data for a in ctx._keep)
    po.L.oracle_lstp_aprod.restype = None
    po.L.oracle_lstp_aprod.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return ctx


def test_the_18_problem_suite_on_the_gpu_matches_the_oracle_and_its_log_parses():
    """lsqr_test (test/lsqrtest_module.f90:55-94) on the device operator.  These problems are
    run to eps-level tolerances with condition numbers up to 6e9, where two compilers of the
    reference itself differ by 0-31 iterations (tests/test_lstp_oracle.py): istop, the
    success pattern and xcheck's verdict must agree, itn within that spread, and x within the
    two runs' own reported errors."""
    buf = io.StringIO()
    res = run_suite(buf)
    parsed = oracle.parse_lis(buf.getvalue())
    assert len(res) == len(parsed) == 18
    po = oracle.port()
    band = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lstp_itn_band.json")))
    for k, ((m, n, nd, p, damp), r, q) in enumerate(zip(SUITE, res, parsed)):
        o = po.lstp_test(m, n, nd, p, damp)
        assert r["istop"] == o["istop"] == 3, k
        assert r["acheck_inform"] == 0 and r["acheck_err"] < 1e-14
        assert r["xcheck_inform"] == o["xcheck_inform"], k
        assert r["success"] == (o["enorm"] <= 1e-3) == (k not in (4, 5)), k
        # 25 singular values repeated 40 times: exact arithmetic would finish in 25 iterations and
        # rounding decides how many more it takes.  tests/golden/lstp_itn_band.json holds the counts of the
        # pinned CPU restatement under six legal evaluation orders of its sums (gen_lstp_band.py): they
        # spread by up to 30 % (325 ... 424), and with pairwise dot products AND pairwise norms -- what a
        # GPU does -- they land within a few iterations of the device's (65/66 vs 66, 140 vs 139, 209/208 vs
        # 208): the device must fall inside that measured spread, widened by its own width.
        itns = list(band[k]["itn"].values())
        assert (band[k]["m"], band[k]["n"], band[k]["npower"]) == (m, n, p) and itns[0] == o["itn"]
        width = max(3, max(itns) - min(itns))
        assert min(itns) - width <= r["itn"] <= max(itns) + width, (k, r["itn"], itns)
        # ... and close to the two variants that sum like a GPU (pairwise dots and norms)
        like_gpu = itns[4:6]
        assert min(like_gpu) * 0.88 - 3 <= r["itn"] <= max(like_gpu) * 1.12 + 3, (k, r["itn"], like_gpu)
        assert r["anorm"] == pytest.approx(o["anorm"], rel=0.15)     # an estimate that grows with itn
        bound = (r["enorm"] + o["enorm"]) * (1.0 + np.linalg.norm(o["xtrue"]))
        assert np.linalg.norm(r["x"] - o["x"]) <= 2 * bound + 1e-12, k
        if r["success"]:
            assert r["enorm"] <= 50 * o["enorm"] + 1e-13, (k, r["enorm"], o["enorm"])
        # the log carries the same facts (layout of the reference's LSQR.LIS)
        assert (q["m"], q["n"], q["npower"], q["istop"], q["itn"], q["xcheck_inform"], q["success"]) == \
               (m, n, p, r["istop"], r["itn"], r["xcheck_inform"], r["success"])
        assert q["anorm"] == pytest.approx(r["anorm"], rel=1e-5) and q["enorm"] == pytest.approx(r["enorm"], rel=1e-2)
        np.testing.assert_allclose(q["x8"], r["x"][:8], rtol=1e-5, atol=1e-12)


class _WrappedMatrix(lsqr_solver_device):
    """A user operator that happens to be an EZ matrix: aprod_device forwards to its device
    product on the SAME stream (what a Fortran/C user would do with their own kernels)."""

    def __init__(self, ez):
        super().__init__()
        self.ez = ez
        self.calls = 0

    def aprod_device(self, mode, m, n, d_x, d_y, stream):
        self.calls += 1
        check(lib().lsqrhip_set_stream(self.ez._h, stream))
        check(lib().lsqrhip_aprod_device(self.ez._h, mode, d_x, d_y))
        return 0


@pytest.mark.parametrize("damp,wantse", [(0.0, False), (1e-2, True)])
def test_user_operator_hook_reproduces_the_matrix_path(damp, wantse):
    p = P.random_rows(3000, 800, 9, seed=21, damp=damp)
    ez = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=1e-10, btol=1e-10, itnlim=200)
    r_ez = ez.solve(p.b, damp, wantse=wantse)
    op = _WrappedMatrix(ez).initialize(p.m, p.n, atol=1e-10, btol=1e-10, itnlim=200)
    r = op.solve(p.b, damp, wantse=wantse)
    assert op.calls >= 2 * r.itn + 1
    assert (r.istop, r.itn) == (r_ez.istop, r_ez.itn)
    assert np.linalg.norm(r.x - r_ez.x) <= 1e-10 * np.linalg.norm(r_ez.x)
    for k in ("anorm", "acond", "rnorm", "xnorm"):
        assert getattr(r, k) == pytest.approx(getattr(r_ez, k), rel=1e-10)
    if wantse:
        assert np.linalg.norm(r.se - r_ez.se) <= 1e-9 * np.linalg.norm(r_ez.se)
    # the reference-style call and the diagnostics on the operator handle
    r2 = op.lsqr(p.m, p.n, damp, wantse, p.b, 1e-10, 1e-10, 0.0, 200)
    assert np.array_equal(r2.x, r.x) and r2.itn == r.itn          # deterministic
    assert op.acheck()[0] == 0
    inform, tests, *_ = op.xcheck(r.anorm, damp, p.b, r.x)
    assert inform in (1, 2, 3)
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=damp, atol=1e-10, btol=1e-10, itnlim=200)
    assert (r.istop, r.itn) == (o.istop, o.itn) and np.linalg.norm(r.x - o.x) <= 1e-10 * np.linalg.norm(o.x)


def test_operator_edge_cases():
    # b = 0: x = 0 is the exact solution, no product after the first (src/lsqr.f90:646-653)
    s = saunders_problem(50, 30, 5, 2, 0.0)
    r = s.solve(np.zeros(50), 0.0)
    assert r.istop == 0 and r.itn == 0 and np.all(r.x == 0.0)
    # itnlim = 1
    s.itnlim = 1
    r = s.solve(s.b, 0.0)
    assert (r.istop, r.itn) == (5, 1)
    # a failing callback surfaces as an error, not a hang
    class Bad(lsqr_solver_device):
        def aprod_device(self, *a):
            raise RuntimeError("boom")
    bad = Bad().initialize(10, 10)
    with pytest.raises(capi.LsqrHipError):
        bad.solve(np.ones(10), 0.0)
    # matrix-only entry points refuse an operator handle
    with pytest.raises(capi.LsqrHipError):
        s.bench_kernel(1, 3)


# ---------------------------------------------------------------------------------------------
# REAL32: the reference's precision macro applies to the abstract class too (src/lsqr_kinds.F90:16-17,
# src/lsqr.f90:16-30) -- lsqrhip_create_operator_f32 / lsqrhip_lstp_create_f32
# ---------------------------------------------------------------------------------------------
def test_real32_suite_on_the_device_operator_follows_the_references_real32_build():
    """The 18 problems on the device operator in REAL32 (problem generated in binary32 arithmetic, real32 vectors on
    the device, binary64 registers) against the UNMODIFIED reference compiled with -DREAL32 running its own
    test/lsqrtest.f90 (oracle/_ref/lsqrtest32 -> tests/golden/real32_lstp_ref.json, gen_real32_golden.py).  In real32
    sixteen of the eighteen "fail" the test's 1e-3 criterion in the reference itself -- their condition numbers
    (1e3 .. 6e9) leave no digits -- so the check is that the device path fails and succeeds WITH it: the same istop,
    the same verdict, the error in x within 3 % of the reference's where that error is the problem's own, and no
    more iterations than the reference needed (binary64 registers lose less per step: it needs 10-40 % fewer)."""
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real32_lstp_ref.json")))
    res = run_suite(real32=True)
    assert len(res) == len(ref) == 18
    for r, g in zip(res, ref):
        assert (r["m"], r["n"], r["npower"]) == (g["m"], g["n"], g["npower"])
        assert r["x"].dtype == np.float32
        assert r["istop"] == g["istop"], (r["npower"], r["istop"], g["istop"])
        assert r["success"] == g["success"]
        # acheck and xcheck in real32 on the device (lsqrhip_acheck_f32 / _xcheck_f32): the reference's verdicts
        assert (r["acheck_inform"] == 0) == g["acheck_ok"] and r["acheck_err"] < 1e-5
        assert r["xcheck_inform"] == g["xcheck_inform"], (r["xcheck_inform"], g["xcheck_inform"])
        if g["success"]:
            assert r["enorm"] <= 1e-3
        else:
            assert abs(r["enorm"] - g["enorm"]) <= 0.03 * g["enorm"], (r["enorm"], g["enorm"])
        assert 0.5 * g["itn"] <= r["itn"] <= 1.05 * g["itn"], (r["itn"], g["itn"])


class _WrappedMatrix32(lsqr_solver_device):
    """A REAL32 matrix handle's own product as a user operator: float vectors in, float vectors out."""

    def __init__(self, ez):
        super().__init__()
        self.ez = ez
        self.calls = 0

    def aprod_device(self, mode, m, n, d_x, d_y, stream):
        self.calls += 1
        check(lib().lsqrhip_set_stream(self.ez._h, stream))
        check(lib().lsqrhip_aprod_device_f32(self.ez._h, mode, d_x, d_y))
        return 0


def test_real32_user_operator_hook_reproduces_the_real32_matrix_path():
    p = P.random_rows(3000, 800, 9, seed=21, damp=1e-2)
    a32, b32 = p.a.astype(np.float32), p.b.astype(np.float32)
    ez = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, atol=1e-6, btol=1e-6, itnlim=200, real32=True)
    r_ez = ez.solve(b32, 1e-2, wantse=True)
    op = _WrappedMatrix32(ez).initialize(p.m, p.n, atol=1e-6, btol=1e-6, itnlim=200, real32=True)
    r = op.solve(b32, 1e-2, wantse=True)
    assert r.x.dtype == np.float32 and op.calls >= 2 * r.itn + 1
    # same real32 vectors, same binary64 arithmetic between them; the norms are summed in another order
    assert r.istop == r_ez.istop and abs(r.itn - r_ez.itn) <= 1
    assert np.linalg.norm(r.x.astype(np.float64) - r_ez.x) <= 2e-5 * np.linalg.norm(r_ez.x)
    assert r.anorm == pytest.approx(r_ez.anorm, rel=1e-5) and r.rnorm == pytest.approx(r_ez.rnorm, rel=1e-5)
    assert op.acheck()[0] == 0 and ez.acheck()[0] == 0               # (real32: lsqrhip_acheck_f32, matrix and operator handles)
    inform, tests, u, v, w = op.xcheck(r.anorm, 1e-2, b32, r.x)
    inform_ez, tests_ez, *_ = ez.xcheck(r_ez.anorm, 1e-2, b32, r_ez.x)
    assert inform in (1, 2, 3) and inform == inform_ez and u.dtype == np.float32
    assert tests[2] == pytest.approx(tests_ez[2], rel=1e-2, abs=1e-7)
    # the binary64 entry points refuse the handle instead of reading floats as doubles
    with pytest.raises(capi.LsqrHipError):
        lsqr_solver_ez.solve(_as64(op), p.b, 1e-2)


def _as64(s):
    class V:            # the same handle presented as a binary64 solver
        pass
    v = lsqr_solver_ez()
    v._h, v.m, v.n, v.real32 = s._h, s.m, s.n, False
    v.atol, v.btol, v.conlim, v.itnlim, v.nout = s.atol, s.btol, s.conlim, s.itnlim, 0
    v._free = lambda: None
    return v
