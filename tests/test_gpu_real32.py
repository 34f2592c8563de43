"""The REAL32 build (src/lsqr_kinds.F90:16-17 of the reference: wp = real32) through the C-ABI:
lsqrhip_create_f32 / lsqrhip_solve_f32 / lsqrhip_aprod_f32 and their device-pointer forms.

All-real32 storage on the device (values, u, v, w, x, se) with binary64 arithmetic in registers.
The checker is the binary64 oracle on the same real32-valued inputs: one product differs from it by
the real32 rounding of the result only; a solve drifts from it by what rounding every vector to
real32 once per iteration does (the reference's own all-real32 iteration rounds every operation,
tests/golden/real32_ref.json).  LSQRHIP_REAL32_MIXED=1 keeps binary64 on the device: results are then
the binary64 ones rounded once."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from cases import build_cases
from lsqr_amd import capi
from lsqr_amd import problems as P
from lsqr_amd.capi import LsqrHipError
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu
CASES = build_cases()
EPS32 = float(np.finfo(np.float32).eps)          # 1.19e-7


@pytest.fixture
def env():
    keys = ("LSQRHIP_CSB", "LSQRHIP_REAL32_MIXED", "LSQRHIP_XLDS", "LSQRHIP_SELL", "LSQRHIP_PAT", "LSQRHIP_SPAT")
    old = {k: os.environ.get(k) for k in keys}

    def put(**kw):
        for k, v in kw.items():
            os.environ[k] = str(v)
    yield put
    for k, v in old.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def as32(p):
    a = np.asarray(p.a, dtype=np.float32)
    b = np.asarray(p.b, dtype=np.float32)
    return a, b


def vecs32(p):
    xp = P.u64_to_unit(P.rng_u64(201, 9, np.arange(p.n, dtype=np.uint64))).astype(np.float32)
    yp = P.u64_to_unit(P.rng_u64(202, 9, np.arange(p.m, dtype=np.uint64))).astype(np.float32)
    return xp, yp


def layouts(put, which):
    if which == "csb":
        put(LSQRHIP_CSB=1)
    elif which == "windows":
        put(LSQRHIP_CSB=0, LSQRHIP_SELL=0, LSQRHIP_PAT=0, LSQRHIP_SPAT=0)
    elif which == "sell":
        put(LSQRHIP_CSB=0, LSQRHIP_PAT=0, LSQRHIP_SPAT=0)
    elif which == "spat":
        put(LSQRHIP_CSB=0, LSQRHIP_PAT=0, LSQRHIP_SPAT=1)
    else:
        put(LSQRHIP_CSB=0)


NAMES = ["random_over_damped", "random_under", "shuffled_dups", "powerlaw_small", "empty_rows_cols",
         "poisson_20x20_it50", "t1_readme_damped", "one_by_one", "zero_matrix", "b_zero"]


@pytest.mark.parametrize("layout", ["default", "csb", "windows", "sell", "spat"])
@pytest.mark.parametrize("name", NAMES)
def test_products_round_once(env, name, layout):
    """y + A x and x + A' y: binary64 sums of exact products of real32 numbers, rounded to real32 once."""
    layouts(env, layout)
    p, o = CASES[name]
    a32, _ = as32(p)
    s = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, real32=True)
    if layout == "csb":
        assert s.info()["xlds"] == 3 and s.info()["xlds_t"] == 3
    xp, yp = vecs32(p)
    po = oracle.port()
    a64 = a32.astype(np.float64)
    y1 = po.aprod(1, p.m, p.n, p.irow, p.icol, a64, xp.astype(np.float64), yp.astype(np.float64))[1]
    x2 = po.aprod(2, p.m, p.n, p.irow, p.icol, a64, xp.astype(np.float64), yp.astype(np.float64))[0]
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    assert np.array_equal(x, xp)
    # |result - exact| <= half an ulp of the result + the binary64 summation error (<< that)
    slack = lambda v: 1e-11 * (1.0 + np.abs(v).max(initial=0.0))
    assert np.all(np.abs(y.astype(np.float64) - y1) <= 0.5 * EPS32 * np.abs(y1) + slack(y1))
    x, y = xp.copy(), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    assert np.array_equal(y, yp)
    assert np.all(np.abs(x.astype(np.float64) - x2) <= 0.5 * EPS32 * np.abs(x2) + slack(x2))


@pytest.mark.parametrize("layout", ["default", "csb", "windows", "sell", "spat"])
@pytest.mark.parametrize("name", NAMES + ["itnlim_1", "illcond_conlim_it10"])
def test_solve_all_real32_follows_binary64_oracle(env, name, layout):
    layouts(env, layout)
    p, o = CASES[name]
    a32, b32 = as32(p)
    itn = min(o["itnlim"], 8 if name == "powerlaw_small" else 25)
    opts = dict(atol=o["atol"], btol=o["btol"], conlim=o["conlim"], itnlim=itn)
    s = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, real32=True, **opts)
    r = s.solve(b32, o["damp"], wantse=o["wantse"])
    assert r.x.dtype == np.float32
    g = oracle.port().solve(p.m, p.n, p.irow, p.icol, a32.astype(np.float64), b32.astype(np.float64),
                            damp=o["damp"], wantse=o["wantse"], **opts)
    # The "1 + test <= 1" stops are taken in real32 like the reference's REAL32 build does (scalar.h), so a
    # real32 run may stop at eps(real32) where the binary64 one goes on; otherwise they stop together.
    assert r.itn <= g.itn + 2
    if r.itn == g.itn:
        assert r.istop == g.istop
    nx = np.linalg.norm(g.x)
    # every vector is rounded to real32 once per iteration: drift ~ itn * eps32 * (conditioning)
    assert np.linalg.norm(r.x - g.x) <= 2e-4 * nx + 1e-30
    if g.itn == r.itn and g.itn > 0:
        for k in ("anorm", "rnorm", "xnorm"):
            assert abs(getattr(r, k) - getattr(g, k)) <= 2e-4 * abs(getattr(g, k)) + 1e-30, k
        if o["wantse"]:
            assert np.linalg.norm(r.se - g.se) <= 2e-3 * np.linalg.norm(g.se) + 1e-30
    # repeats itself exactly
    r2 = s.solve(b32, o["damp"], wantse=o["wantse"])
    assert np.array_equal(r2.x, r.x) and (r2.itn, r2.anorm, r2.rnorm) == (r.itn, r.anorm, r.rnorm)


@pytest.mark.parametrize("name", ["random_over_damped", "poisson_20x20_it50", "shuffled_dups"])
def test_mixed_mode_is_binary64_rounded_once(env, name):
    env(LSQRHIP_REAL32_MIXED=1)
    p, o = CASES[name]
    a32, b32 = as32(p)
    opts = dict(atol=o["atol"], btol=o["btol"], conlim=o["conlim"], itnlim=min(o["itnlim"], 25))
    s = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, real32=True, **opts)
    r = s.solve(b32, o["damp"], wantse=True)
    g = oracle.port().solve(p.m, p.n, p.irow, p.icol, a32.astype(np.float64), b32.astype(np.float64),
                            damp=o["damp"], wantse=True, **opts)
    assert (r.istop, r.itn) == (g.istop, g.itn)
    assert np.max(np.abs(r.x - g.x)) <= 0.6 * EPS32 * np.max(np.abs(g.x)) + 1e-9 * np.max(np.abs(g.x))
    assert abs(r.rnorm - g.rnorm) <= 1e-9 * g.rnorm
    xp, yp = vecs32(p)
    y = yp.copy()
    s.aprod(1, p.m, p.n, xp.copy(), y)
    y1 = oracle.port().aprod(1, p.m, p.n, p.irow, p.icol, a32.astype(np.float64), xp.astype(np.float64),
                             yp.astype(np.float64))[1]
    assert np.all(np.abs(y - y1) <= 0.5 * EPS32 * np.abs(y1) + 1e-12)


def test_device_pointer_forms_and_refusals(env):
    p, o = CASES["random_over_damped"]
    a32, b32 = as32(p)
    s = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, real32=True, itnlim=20)
    r = s.solve(b32, o["damp"], wantse=True)
    L = capi.lib()
    d_b = capi.DeviceBuffer.from_array(b32)
    d_x = capi.DeviceBuffer(4 * p.n)
    d_se = capi.DeviceBuffer(4 * p.n)
    istop, itn = C.c_int(), C.c_int()
    sc = [C.c_double() for _ in range(5)]
    capi.check(L.lsqrhip_solve_device_f32(s._h, d_b.ptr.value, o["damp"], 0.0, 0.0, 0.0, 20, 1, 0, d_x.ptr.value,
                                          d_se.ptr.value, C.addressof(istop), C.addressof(itn),
                                          *[C.addressof(v) for v in sc]))
    assert (istop.value, itn.value) == (r.istop, r.itn)
    assert np.array_equal(d_x.to_array(np.float32, p.n), r.x)
    assert np.array_equal(d_se.to_array(np.float32, p.n), r.se)
    xp, yp = vecs32(p)
    dx, dy = capi.DeviceBuffer.from_array(xp), capi.DeviceBuffer.from_array(yp)
    capi.check(L.lsqrhip_aprod_device_f32(s._h, 1, dx.ptr.value, dy.ptr.value))
    y = yp.copy()
    s.aprod(1, p.m, p.n, xp.copy(), y)
    assert np.array_equal(dy.to_array(np.float32, p.m), y)
    # the binary64 entry points refuse a REAL32 handle instead of misreading its arrays
    b64, x64 = np.zeros(p.m), np.zeros(p.n)
    rc = L.lsqrhip_solve(s._h, b64.ctypes.data, 0.0, 0.0, 0.0, 0.0, 5, 0, 0, x64.ctypes.data, None,
                         C.addressof(istop), C.addressof(itn), *[C.addressof(v) for v in sc])
    assert rc == capi.ERR_ARG
    assert L.lsqrhip_aprod(s._h, 1, x64.ctypes.data, b64.ctypes.data) == capi.ERR_ARG
    assert L.lsqrhip_solve_device(s._h, d_b.ptr.value, 0.0, 0.0, 0.0, 0.0, 5, 0, 0, d_x.ptr.value, None,
                                  C.addressof(istop), C.addressof(itn), *[C.addressof(v) for v in sc]) == capi.ERR_ARG
    inform, err = C.c_int(), C.c_double()
    assert L.lsqrhip_acheck(s._h, 1e-16, C.addressof(inform), C.addressof(err)) == capi.ERR_ARG
    # ... and the real32 ones refuse a binary64 handle
    s64 = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    assert L.lsqrhip_aprod_f32(s64._h, 1, xp.ctypes.data, yp.ctypes.data) == capi.ERR_ARG
    assert L.lsqrhip_aprod_device_f32(s64._h, 1, dx.ptr.value, dy.ptr.value) == capi.ERR_ARG
    assert L.lsqrhip_acheck_f32(s64._h, 1e-7, C.addressof(inform), C.addressof(err)) == capi.ERR_ARG
    # acheck on the REAL32 handle itself (round 4: lsqrhip_acheck_f32 -- real32 vectors, the adjoint identity to real32)
    ainf, aerr = s.acheck()
    assert ainf == 0 and aerr < 1e-6


def test_log_records_of_a_real32_solve(env):
    p, o = CASES["poisson_20x20_it50"]
    a32, b32 = as32(p)
    s = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, real32=True, itnlim=30)
    L = capi.lib()
    x = np.zeros(p.n, dtype=np.float32)
    istop, itn = C.c_int(), C.c_int()
    sc = [C.c_double() for _ in range(5)]
    capi.check(L.lsqrhip_solve_f32(s._h, b32.ctypes.data, 0.0, 0.0, 0.0, 0.0, 30, 0, 1, x.ctypes.data, None,
                                   C.addressof(istop), C.addressof(itn), *[C.addressof(v) for v in sc]))
    rec = s.log_records()
    # (records follow the reference's selective-print rule: the first and the last ten iterations here)
    assert rec.shape[0] == 21 and rec[0, 0] == 1 and rec[-1, 0] == itn.value
    assert abs(rec[-1, 1] - float(x[0])) <= EPS32 * abs(float(x[0]))   # x(1): the float value the device holds


@pytest.mark.parametrize("ngpu", [1, 3])
@pytest.mark.parametrize("name", ["random_over_damped", "random_over_se", "poisson_20x20_it50", "shuffled_dups",
                                  "empty_rows_cols", "b_zero", "one_by_one"])
def test_sharded_real32_handle(env, name, ngpu):
    """lsqrhip_create_sharded_f32 (what `initialize(..., ngpu=)` of the -DREAL32 host layer binds): real32 storage
    in every row block and real32 slices in both exchanges, binary64 registers -- through lsqrhip_solve_f32 /
    lsqrhip_aprod_f32 with float host vectors, several ranks on this GPU (loopback).  Same bounds as the one-GPU
    REAL32 handle: a product is off by the real32 rounding of the result (of each rank's partial result in mode
    2), a solve follows the binary64 oracle to what rounding every vector once per iteration costs."""
    os.environ["LSQRHIP_SHARD_LOOPBACK"] = "1"
    try:
        p, o = CASES[name]
        a32, b32 = as32(p)
        L = capi.lib()
        h = C.c_void_p()
        irow = np.ascontiguousarray(p.irow, np.int32)
        icol = np.ascontiguousarray(p.icol, np.int32)
        capi.check(L.lsqrhip_create_sharded_f32(p.m, p.n, a32.size, irow.ctypes.data, icol.ctypes.data, a32.ctypes.data,
                                                ngpu, C.byref(h)))
        try:
            itn_lim = min(o["itnlim"], 25)
            x, se = np.zeros(max(p.n, 1), np.float32), np.zeros(max(p.n, 1), np.float32)
            istop, itn = C.c_int(), C.c_int()
            sc = [C.c_double() for _ in range(5)]

            def solve():
                capi.check(L.lsqrhip_solve_f32(h, b32.ctypes.data, o["damp"], o["atol"], o["btol"], o["conlim"], itn_lim,
                                               int(o["wantse"]), 0, x.ctypes.data, se.ctypes.data if o["wantse"] else None,
                                               C.addressof(istop), C.addressof(itn), *[C.addressof(v) for v in sc]))
                return x[:p.n].copy(), istop.value, itn.value, [v.value for v in sc]
            xs, is_, it_, scal = solve()
            g = oracle.port().solve(p.m, p.n, p.irow, p.icol, a32.astype(np.float64), b32.astype(np.float64),
                                    damp=o["damp"], wantse=o["wantse"], atol=o["atol"], btol=o["btol"],
                                    conlim=o["conlim"], itnlim=itn_lim)
            assert it_ <= g.itn + 2
            if it_ == g.itn:
                assert is_ == g.istop
            nx = np.linalg.norm(g.x)
            assert np.linalg.norm(xs - g.x) <= 2e-4 * nx + 1e-30
            if it_ == g.itn and g.itn > 0:
                assert abs(scal[0] - g.anorm) <= 2e-4 * g.anorm and abs(scal[2] - g.rnorm) <= 2e-4 * g.rnorm + 1e-30
                if o["wantse"]:
                    assert np.linalg.norm(se[:p.n] - g.se) <= 2e-3 * np.linalg.norm(g.se) + 1e-30
            xs2, is2, it2, scal2 = solve()               # repeats itself exactly
            assert np.array_equal(xs2, xs) and (is2, it2, scal2) == (is_, it_, scal)
            if p.nnz:
                xp, yp = vecs32(p)
                a64 = a32.astype(np.float64)
                po = oracle.port()
                y1 = po.aprod(1, p.m, p.n, p.irow, p.icol, a64, xp.astype(np.float64), yp.astype(np.float64))[1]
                x2 = po.aprod(2, p.m, p.n, p.irow, p.icol, a64, xp.astype(np.float64), yp.astype(np.float64))[0]
                xx, yy = xp.copy(), yp.copy()
                capi.check(L.lsqrhip_aprod_f32(h, 1, xx.ctypes.data, yy.ctypes.data))
                assert np.all(np.abs(yy.astype(np.float64) - y1) <= 0.5 * EPS32 * np.abs(y1) + 1e-11 * (1 + np.abs(y1).max()))
                xx, yy = xp.copy(), yp.copy()
                capi.check(L.lsqrhip_aprod_f32(h, 2, xx.ctypes.data, yy.ctypes.data))
                assert np.all(np.abs(xx.astype(np.float64) - x2) <= (ngpu + 1) * EPS32 * np.abs(x2).max() + 1e-11)
            # the binary64 entry points refuse it
            b64, x64 = np.zeros(max(p.m, 1)), np.zeros(max(p.n, 1))
            rc = L.lsqrhip_solve(h, b64.ctypes.data, 0.0, 0.0, 0.0, 0.0, 5, 0, 0, x64.ctypes.data, None,
                                 C.addressof(istop), C.addressof(itn), *[C.addressof(v) for v in sc])
            assert rc == capi.ERR_ARG
        finally:
            capi.check(L.lsqrhip_destroy(h))
    finally:
        os.environ.pop("LSQRHIP_SHARD_LOOPBACK", None)


@pytest.mark.parametrize("ngpu,csb,parts", [(3, None, 2), (8, "1", 2), (4, "1", 4)])
def test_sharded_real32_with_overlapped_exchanges_changes_no_bit(env, ngpu, csb, parts):
    """LSQRHIP_SHARD_OVERLAP=1 on REAL32 row blocks (real32 slices in parts on the exchange stream; column-swept
    float layouts built for the parts): the solve of the plain schedule, bit for bit (loopback harness)."""
    keys = ("LSQRHIP_SHARD_LOOPBACK", "LSQRHIP_CSB", "LSQRHIP_SHARD_OVERLAP", "LSQRHIP_SHARD_PARTS")
    old = {k: os.environ.get(k) for k in keys}
    p = P.random_rows(40000, 9001, 10, seed=37, damp=1e-2)
    a32, b32 = p.a.astype(np.float32), p.b.astype(np.float32)
    L = capi.lib()
    irow, icol = np.ascontiguousarray(p.irow, np.int32), np.ascontiguousarray(p.icol, np.int32)
    out = []
    try:
        os.environ["LSQRHIP_SHARD_LOOPBACK"] = "1"
        os.environ["LSQRHIP_SHARD_PARTS"] = str(parts)
        if csb:
            os.environ["LSQRHIP_CSB"] = csb
        for overlap in ("0", "1"):
            os.environ["LSQRHIP_SHARD_OVERLAP"] = overlap
            h = C.c_void_p()
            capi.check(L.lsqrhip_create_sharded_f32(p.m, p.n, a32.size, irow.ctypes.data, icol.ctypes.data,
                                                    a32.ctypes.data, ngpu, C.byref(h)))
            try:
                x, se = np.zeros(p.n, np.float32), np.zeros(p.n, np.float32)
                istop, itn = C.c_int(), C.c_int()
                sc = [C.c_double() for _ in range(5)]
                capi.check(L.lsqrhip_solve_f32(h, b32.ctypes.data, 1e-2, 1e-6, 1e-6, 0.0, 30, 1, 0, x.ctypes.data,
                                               se.ctypes.data, C.addressof(istop), C.addressof(itn),
                                               *[C.addressof(v) for v in sc]))
                out.append((x.copy(), se.copy(), istop.value, itn.value, [v.value for v in sc]))
            finally:
                capi.check(L.lsqrhip_destroy(h))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    off, on = out
    assert off[2:] == on[2:] and off[3] > 3
    assert np.array_equal(off[0], on[0]) and np.array_equal(off[1], on[1])
