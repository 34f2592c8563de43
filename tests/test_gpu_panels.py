"""Column-panel SpMV layout (spmv.h "Column panels"): forced on at test scale with a small
panel size, checked against the oracle for both aprod modes, a full solve, the sharded-stage
form, and run-to-run determinism."""
import os

import numpy as np
import pytest

import oracle
from lsqr_amd import problems as P
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced_panels():
    old = {k: os.environ.get(k) for k in ("LSQRHIP_PANELS", "LSQRHIP_PANEL_KB", "LSQRHIP_CSB")}
    os.environ["LSQRHIP_PANELS"] = "1"
    os.environ["LSQRHIP_PANEL_KB"] = "64"      # 8192 columns per panel
    os.environ["LSQRHIP_CSB"] = "0"            # the panel kernels themselves (csb.h replaces them by default)
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("shape", [(30000, 40000, 6), (50000, 20000, 9), (20000, 60000, 30)])
def test_panelled_aprod_and_solve_match_oracle(forced_panels, shape):
    m, n, per = shape
    p = P.random_rows(m, n, per, seed=5, damp=1e-3)
    # 12 iterations: panel mode changes the order of the row sums (per-panel segments), and LSQR
    # amplifies such ulp-level differences with the iteration count -- the
    # reference drifts the same way under a permutation of its COO input
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=12)
    po = oracle.port()
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(y - y_ref)) <= 1e-13 * max(np.max(np.abs(y_ref)), 1.0)
    x, y = xp.copy(), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(x - x_ref)) <= 1e-13 * max(np.max(np.abs(x_ref)), 1.0)
    r = s.solve(p.b, 1e-3)
    o = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=1e-3, itnlim=12)
    assert (r.istop, r.itn) == (o.istop, o.itn)
    assert np.linalg.norm(r.x - o.x) <= 1e-10 * np.linalg.norm(o.x)
    assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * o.rnorm
    r2 = s.solve(p.b, 1e-3)
    assert np.array_equal(r.x, r2.x) and r.anorm == r2.anorm          # deterministic
    s.set_option("pipeline", 0)
    r3 = s.solve(p.b, 1e-3)
    assert np.array_equal(r.x, r3.x) and r.anorm == r3.anorm          # both schedules, same bits


def test_skewed_windows_set_long_segments_aside(forced_panels):
    """Power-law rows cut by 4 panels: windows whose mean segment is ~5 hold segments of 130-1000
    (and a few >= 1024 for phase 3).  The build flags those windows and a whole wave sums each
    long segment after phase 2 (spmv.h "phase 2b").  Against the oracle to rounding, identical
    from run to run and across schedules, and agreeing with the unflagged path (LSQRHIP_SKEW=0:
    every segment on the lanes its window's mean calls for) to rounding."""
    p = P.powerlaw_rows(6000, 30000, seed=11, dmin=4, dmax=5000, gamma=1.6, damp=1e-3)
    deg = np.bincount(p.irow - 1, minlength=p.m)
    assert (deg > 4 * 140).sum() > 20 and deg.max() >= 4096        # long segments and a phase-3 row
    po = oracle.port()
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    o = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=1e-3, itnlim=10)
    res = {}
    for skew in ("1", "0"):
        os.environ["LSQRHIP_SKEW"] = skew
        os.environ["LSQRHIP_XLDS"] = "0"            # L2 panels (rows this dense would go to LDS panels)
        try:
            s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=10)
            assert s.info()["panels"] == 4 and s.info()["xlds"] == 0
            x, y = xp.copy(), yp.copy()
            s.aprod(1, p.m, p.n, x, y)
            assert np.max(np.abs(y - y_ref)) <= 1e-13 * np.max(np.abs(y_ref))
            y1 = y.copy()
            x, y = xp.copy(), yp.copy()
            s.aprod(1, p.m, p.n, x, y)
            assert np.array_equal(y, y1)                                 # which wave takes which segment never shows
            x, y = xp.copy(), yp.copy()
            s.aprod(2, p.m, p.n, x, y)
            assert np.max(np.abs(x - x_ref)) <= 1e-13 * np.max(np.abs(x_ref))
            out = []
            for pipeline in (0, 1, 2):
                s.set_option("pipeline", pipeline)
                r = s.solve(p.b, 1e-3)
                assert (r.istop, r.itn) == (o.istop, o.itn)
                assert np.linalg.norm(r.x - o.x) <= 1e-9 * np.linalg.norm(o.x)
                out.append(r)
            assert all(np.array_equal(out[0].x, r.x) and out[0].anorm == r.anorm for r in out[1:])
            res[skew] = (y1, out[0])
        finally:
            os.environ.pop("LSQRHIP_SKEW", None)
            os.environ.pop("LSQRHIP_XLDS", None)
    assert not np.array_equal(res["1"][0], res["0"][0])                   # the flagged path really ran
    assert np.max(np.abs(res["1"][0] - res["0"][0])) <= 1e-13 * np.max(np.abs(y_ref))
    assert np.linalg.norm(res["1"][1].x - res["0"][1].x) <= 1e-9 * np.linalg.norm(o.x)


def test_panels_are_not_used_for_banded_matrices():
    """Auto mode keeps the plain CSR for a banded system even when x exceeds L2: the aprod
    result stays bit-identical to the reference's row sums (only the plain path is)."""
    for k in ("LSQRHIP_PANELS", "LSQRHIP_PANEL_KB"):
        os.environ.pop(k, None)
    p = P.poisson2d(1200, 1000)        # n = 1.2e6: 9.6 MB of x
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    y = np.zeros(p.m)
    s.aprod(1, p.m, p.n, xp, y)
    _, y_ref = oracle.port().aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, np.zeros(p.m))
    assert np.array_equal(y, y_ref)


def test_wide_rows_keep_32bit_columns_and_match_oracle():
    """Row blocks spanning >= 65536 columns cannot use the 16-bit block-relative column
    indices (spmv.h C16): the 32-bit plain-CSR path must give the same answers."""
    for k in ("LSQRHIP_PANELS", "LSQRHIP_PANEL_KB", "LSQRHIP_COL16"):
        os.environ.pop(k, None)
    p = P.random_rows(3000, 200000, 12, seed=9, damp=1e-3)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=15)
    po = oracle.port()
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    y = np.zeros(p.m)
    s.aprod(1, p.m, p.n, xp, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, np.zeros(p.m))
    assert np.array_equal(y, y_ref)             # plain path: bit-identical row sums
    r = s.solve(p.b, 1e-3)
    o = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=1e-3, itnlim=15)
    assert (r.istop, r.itn) == (o.istop, o.itn)
    assert np.linalg.norm(r.x - o.x) <= 1e-10 * np.linalg.norm(o.x)


def test_col16_and_col32_paths_agree_bitwise():
    p = P.poisson2d(300, 200)
    res = []
    for flag in ("1", "0"):
        os.environ["LSQRHIP_COL16"] = flag
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=40)
        r = s.solve(p.b, 0.0)
        res.append((r.x.copy(), r.anorm, r.rnorm, r.itn))
    os.environ.pop("LSQRHIP_COL16", None)
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1:] == res[1][1:]


@pytest.fixture
def forced_xlds():
    keys = ("LSQRHIP_XLDS", "LSQRHIP_XLDS_COLS", "LSQRHIP_PANELS", "LSQRHIP_PANEL_KB")
    old = {k: os.environ.get(k) for k in keys}
    os.environ["LSQRHIP_XLDS"] = "1"
    os.environ["LSQRHIP_XLDS_COLS"] = "1024"        # 1024-column panels whose x slice lives in LDS
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("shape", [(3000, 20000, 120), (20000, 3000, 60), (5000, 5000, 9), (2500, 9000, 1500)])
def test_lds_resident_panels_match_oracle(forced_xlds, shape):
    """spmv.h XL: panels narrow enough for the x slice to live in LDS (BASELINE config 3 at its
    literal 1000 per row chooses them by itself); forced here at test scale, including rows long
    enough for the workgroup-split path (1500 per row) and windows that straddle two panels."""
    m, n, per = shape
    p = P.random_rows(m, n, per, seed=6, damp=1e-3)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=12)
    info = s.info()
    assert (info["xlds"] != 0) == (n > 1024) and (info["xlds_t"] != 0) == (m > 1024)
    assert info["panels"] == -(-n // 1024) and info["panels_t"] == -(-m // 1024)
    po = oracle.port()
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(y - y_ref)) <= 1e-13 * max(np.max(np.abs(y_ref)), 1.0)
    x, y = xp.copy(), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(x - x_ref)) <= 1e-13 * max(np.max(np.abs(x_ref)), 1.0)
    r = s.solve(p.b, 1e-3)
    o = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=1e-3, itnlim=12)
    assert (r.istop, r.itn) == (o.istop, o.itn)
    assert np.linalg.norm(r.x - o.x) <= 1e-10 * np.linalg.norm(o.x)
    assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * o.rnorm
    r2 = s.solve(p.b, 1e-3)
    assert np.array_equal(r.x, r2.x) and r.anorm == r2.anorm          # deterministic
    for pipeline in (0, 1):
        s.set_option("pipeline", pipeline)
        r3 = s.solve(p.b, 1e-3)
        assert np.array_equal(r.x, r3.x) and r.anorm == r3.anorm      # every schedule, same bits


@pytest.mark.parametrize("off64", ["0", "1"])
def test_lds_panels_with_dictionary_values_and_64bit_row_pointers(forced_xlds, off64):
    """The LDS-panel kernel's other template variants: one-byte dictionary values (V8) and 64-bit
    row pointers (what BASELINE config 3 uses at its literal 4e9 nonzeros)."""
    os.environ["LSQRHIP_OFF64"] = off64
    saved_val8 = os.environ.pop("LSQRHIP_VAL8", None)      # this test is about the dictionary variant
    try:
        p = P.random_rows(4000, 9000, 40, seed=12, damp=1e-3)
        vals = np.array([0.5, -1.0, 2.0, 0.25, -0.125, 3.0, -0.0])
        a = vals[(P.rng_u64(3, 5, np.arange(p.a.size, dtype=np.uint64)) % np.uint64(vals.size)).astype(np.int64)]
        s = lsqr_solver_ez().initialize(p.m, p.n, a, p.irow, p.icol, itnlim=10)
        info = s.info()
        assert info["xlds"] == 2 and info["xlds_t"] == 2 and info["dict_entries"] == vals.size
        assert info["rowptr_bytes"] == (8 if off64 == "1" else 4) and info["col_bytes"] == 2
        po = oracle.port()
        xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
        yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
        for mode in (1, 2):
            x, y = xp.copy(), yp.copy()
            s.aprod(mode, p.m, p.n, x, y)
            xr, yr = po.aprod(mode, p.m, p.n, p.irow, p.icol, a, xp, yp)
            assert np.max(np.abs(x - xr)) <= 1e-13 * max(np.max(np.abs(xr)), 1.0)
            assert np.max(np.abs(y - yr)) <= 1e-13 * max(np.max(np.abs(yr)), 1.0)
        r = s.solve(p.b, 1e-3)
        o = po.solve(p.m, p.n, p.irow, p.icol, a, p.b, damp=1e-3, itnlim=10)
        assert (r.istop, r.itn) == (o.istop, o.itn) and np.linalg.norm(r.x - o.x) <= 1e-10 * np.linalg.norm(o.x)
    finally:
        os.environ.pop("LSQRHIP_OFF64", None)
        if saved_val8 is not None:
            os.environ["LSQRHIP_VAL8"] = saved_val8
