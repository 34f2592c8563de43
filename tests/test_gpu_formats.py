"""Storage formats behind aprod: the sliced-ELL layout for short even rows (csrc/sell.h), the
one-byte value dictionary (csrc/valdict.h) and 16-bit columns.  None of them may change a
result: aprod must stay bit-identical to the reference's row sums (oracle), and every
combination of formats must agree with every other on a whole solve."""
import itertools
import os

import numpy as np
import pytest

import oracle
from lsqr_amd import problems as P
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu

KNOBS = ("LSQRHIP_SELL", "LSQRHIP_SELLP", "LSQRHIP_VAL8", "LSQRHIP_COL16", "LSQRHIP_PANELS", "LSQRHIP_PANEL_KB", "LSQRHIP_OFF64")


@pytest.fixture(autouse=True)
def clean_env():
    old = {k: os.environ.pop(k, None) for k in KNOBS + ("LSQRHIP_PAT", "LSQRHIP_SPAT")}
    os.environ["LSQRHIP_PAT"] = "0"       # this file is about the layouts UNDER the patterns (tests/test_gpu_patterns.py)
    os.environ["LSQRHIP_SPAT"] = "0"
    yield
    for k, v in old.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def _vec(stream, n):
    return P.u64_to_unit(P.rng_u64(101, stream, np.arange(n, dtype=np.uint64)))


def _banded_real(m, n, seed=3):
    """Tridiagonal-plus-far-diagonal matrix with arbitrary real values (no dictionary), ragged
    at the ends, with a few empty rows."""
    rows, cols = [], []
    for off in (-7, -1, 0, 1, 7):
        r = np.arange(m)
        c = r + off
        ok = (c >= 0) & (c < n) & (r % 97 != 5)      # rows = 5 mod 97 are empty
        rows.append(r[ok]); cols.append(c[ok])
    irow = np.concatenate(rows); icol = np.concatenate(cols)
    order = np.lexsort((icol, irow))
    irow, icol = irow[order], icol[order]
    a = P.u64_to_unit(P.rng_u64(seed, 1, np.arange(irow.size, dtype=np.uint64)))
    b = _vec(7, m)
    return m, n, (irow + 1).astype(np.int32), (icol + 1).astype(np.int32), a, b


def test_default_formats_for_a_stencil():
    p = P.poisson2d(300, 200)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    info = s.info()
    assert info["sell"] == 2 and info["sell_t"] == 2                   # 3-byte nonzeros: packed records
    assert info["dict_entries"] == 2 and info["value_bytes"] == 1     # {4, -1}
    assert info["col_bytes"] == 2 and info["colt_bytes"] == 2
    os.environ["LSQRHIP_SELLP"] = "0"
    info = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()
    assert info["sell"] == 1 and info["sell_t"] == 1 and info["value_bytes"] == 1
    os.environ["LSQRHIP_SELL"] = "0"
    os.environ["LSQRHIP_VAL8"] = "0"
    info = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()
    assert info["sell"] == 0 and info["dict_entries"] == 0 and info["value_bytes"] == 8


def test_irregular_rows_keep_the_window_layout():
    p = P.powerlaw_rows(20000, 30000, dmax=300, seed=2)
    info = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()
    assert info["sell"] == 0 and info["sell_t"] == 0 and info["dict_entries"] == 0
    # rows of one fixed length but scattered columns: only when forced; the ragged transpose never
    p = P.random_rows(20000, 200000, 8, seed=2)
    info = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()
    assert info["sell"] == 0 and info["sell_t"] == 0
    os.environ["LSQRHIP_SELL"] = "1"
    info = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()
    assert info["sell"] == 1 and info["sell_t"] == 0 and info["col_bytes"] == 4


@pytest.mark.parametrize("problem", ["poisson", "banded_real", "poisson_tall", "random_fixed"])
def test_every_format_combination_gives_the_same_bits(problem):
    if problem == "poisson":
        p = P.poisson2d(257, 131)
        m, n, irow, icol, a, b = p.m, p.n, p.irow, p.icol, p.a, p.b
    elif problem == "poisson_tall":
        p = P.poisson2d(64, 3)                       # exactly 3 slices, boundary rows everywhere
        m, n, irow, icol, a, b = p.m, p.n, p.irow, p.icol, p.a, p.b
    elif problem == "random_fixed":                  # 8 per row, scattered columns: 32-bit SELL when forced
        p = P.random_rows(30000, 200000, 8, seed=4)
        m, n, irow, icol, a, b = p.m, p.n, p.irow, p.icol, p.a, p.b
    else:
        m, n, irow, icol, a, b = _banded_real(70001, 70003)
    po = oracle.port()
    xp, yp = _vec(9, n), _vec(10, m)
    _, y_ref = po.aprod(1, m, n, irow, icol, a, xp, yp)
    x_ref, _ = po.aprod(2, m, n, irow, icol, a, xp, yp)
    o = po.solve(m, n, irow, icol, a, b, damp=0.0, itnlim=30)
    results = []
    combos = [c + ("1",) for c in itertools.product("10", repeat=3)] + [("1", "1", "1", "0")]
    for sell, val8, col16, sellp in combos:
        os.environ.update(LSQRHIP_SELL=sell, LSQRHIP_VAL8=val8, LSQRHIP_COL16=col16, LSQRHIP_SELLP=sellp)
        s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=30)
        info = s.info()
        assert (info["sell"] != 0) == (sell == "1")
        packed = sell == val8 == col16 == sellp == "1" and info["dict_entries"] > 0 and info["col_bytes"] == 2
        assert (info["sell"] == 2) == packed, (info, sell, val8, col16, sellp)
        x, y = xp.copy(), yp.copy()
        s.aprod(1, m, n, x, y)
        assert np.array_equal(y, y_ref), (sell, val8, col16)      # the reference's row sums, bit for bit
        x, y = xp.copy(), yp.copy()
        s.aprod(2, m, n, x, y)
        assert np.array_equal(x, x_ref), (sell, val8, col16)
        r = s.solve(b, 0.0)
        assert (r.istop, r.itn) == (o.istop, o.itn)
        assert np.linalg.norm(r.x - o.x) <= 1e-10 * np.linalg.norm(o.x)
        assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * max(o.rnorm, 1e-300)
        results.append((sell, r))
    # the value dictionary and the column width never change a bit within one layout
    for layout in "10":
        rs = [r for s_, r in results if s_ == layout]
        for r in rs[1:]:
            assert np.array_equal(r.x, rs[0].x) and r.anorm == rs[0].anorm and r.rnorm == rs[0].rnorm


def test_dictionary_preserves_signed_zero_and_many_values():
    """200 distinct values including -0.0 and +0.0 (distinct bit patterns): the coded matrix must
    reproduce the 8-byte one exactly; 300 distinct values must fall back to 8-byte storage."""
    m = n = 20000
    for nvals, expect in ((200, 200), (300, 0)):
        table = np.linspace(-3.0, 3.0, nvals)
        table[0], table[1] = -0.0, 0.0
        rows, cols = [], []
        for off in (-2, 0, 3):
            r = np.arange(m); c = r + off
            ok = (c >= 0) & (c < n)
            rows.append(r[ok]); cols.append(c[ok])
        irow = np.concatenate(rows); icol = np.concatenate(cols)
        order = np.lexsort((icol, irow))
        irow, icol = irow[order], icol[order]
        a = table[(P.rng_u64(5, 2, np.arange(irow.size, dtype=np.uint64)) % np.uint64(nvals)).astype(np.int64)]
        irow1, icol1 = (irow + 1).astype(np.int32), (icol + 1).astype(np.int32)
        s = lsqr_solver_ez().initialize(m, n, a, irow1, icol1)
        assert s.info()["dict_entries"] == expect
        xp, yp = _vec(3, n), np.zeros(m)
        x, y = xp.copy(), yp.copy()
        s.aprod(1, m, n, x, y)
        _, y_ref = oracle.port().aprod(1, m, n, irow1, icol1, a, xp, yp)
        assert np.array_equal(y, y_ref) and np.array_equal(np.signbit(y), np.signbit(y_ref))


def test_sell_handles_non_finite_x_like_the_reference():
    """Padding slots are loaded but never added: an inf in x must reach exactly the rows whose
    real entries touch it (0 * inf would poison whole slices otherwise)."""
    p = P.poisson2d(100, 100)
    xp = _vec(4, p.n)
    xp[0] = np.inf            # column 1 is every slice's... smallest column only for slice 0
    xp[5000] = np.inf
    with np.errstate(invalid="ignore"):
        _, y_ref = oracle.port().aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, np.zeros(p.m))
    for sellp in ("1", "0"):
        os.environ["LSQRHIP_SELLP"] = sellp
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
        assert s.info()["sell"] == (2 if sellp == "1" else 1)
        y = np.zeros(p.m)
        x = xp.copy()
        s.aprod(1, p.m, p.n, x, y)
        assert np.array_equal(y, y_ref, equal_nan=True)
        assert np.isfinite(y).sum() == np.isfinite(y_ref).sum() >= p.m - 10


def _ragged_local(m, n, wmax, seed):
    """Rows of 1..wmax nonzeros (the width changes every 128 rows; a few rows are shorter or
    empty), columns within +-40 of the diagonal in a COO order that is NOT column-sorted, values
    from a 7-entry table."""
    rs = np.random.RandomState(seed)
    table = np.array([-2.5, -1.0, -0.0, 0.0, 0.75, 3.0, 1e-3])
    irow, icol = [], []
    for r in range(m):
        cap = 1 + (r // 128) % wmax
        k = 0 if r % 131 == 0 else (rs.randint(0, cap + 1) if r % 17 == 3 else cap)
        c0 = min(max(int(r * (n - 1) / max(m - 1, 1)), 40), n - 41)
        cs = c0 + rs.choice(81, size=k, replace=False) - 40
        irow += [r] * k
        icol += list(cs)
    irow, icol = np.array(irow), np.array(icol)
    a = table[rs.randint(0, table.size, size=irow.size)]
    b = _vec(17, m)
    return m, n, (irow + 1).astype(np.int32), (icol + 1).astype(np.int32), a, b


@pytest.mark.parametrize("shape", [(5000, 4000, 5), (4100, 6000, 13), (3333, 3333, 23)])
def test_packed_records_with_ragged_multi_record_rows(shape):
    """Slices of 1..23 nonzeros per row = 1..5 records per row, empty rows, a last slice that is
    not full: the packed layout (forced) must give the reference's row sums bit for bit in both
    modes and the same solve as the unpacked slices and the row windows, in every schedule."""
    m, n, wmax = shape
    m, n, irow, icol, a, b = _ragged_local(m, n, wmax, seed=wmax)
    po = oracle.port()
    xp, yp = _vec(9, n), _vec(10, m)
    _, y_ref = po.aprod(1, m, n, irow, icol, a, xp, yp)
    x_ref, _ = po.aprod(2, m, n, irow, icol, a, xp, yp)
    o = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=40)
    sols = []
    for sell, sellp, want in (("1", "1", 2), ("1", "0", 1), ("0", "0", 0)):
        os.environ.update(LSQRHIP_SELL=sell, LSQRHIP_SELLP=sellp)
        s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=40)
        info = s.info()
        assert info["sell"] == want, info
        x, y = xp.copy(), yp.copy()
        s.aprod(1, m, n, x, y)
        if want != 0:      # one lane per row: the reference's left-to-right sums
            assert np.array_equal(y, y_ref), (want, int(np.sum(y != y_ref)))
        else:              # row windows split rows longer than 16 over several lanes (spmv.h)
            assert np.max(np.abs(y - y_ref)) <= 1e-14 * np.max(np.abs(y_ref))
        x, y = xp.copy(), yp.copy()
        s.aprod(2, m, n, x, y)
        if info["sell_t"] != 0:
            assert np.array_equal(x, x_ref)
        else:              # (the ragged transpose usually stays in row windows)
            assert np.max(np.abs(x - x_ref)) <= 1e-14 * np.max(np.abs(x_ref))
        for pipeline in (0, 1, 2):
            s.set_option("pipeline", pipeline)
            r = s.solve(b, 1e-2, wantse=True)
            assert (r.istop, r.itn) == (o.istop, o.itn)
            # 40 iterations, far from converged: the tree-summed norms move x by ~1e-10 against
            # the oracle's sequential ones (tests/test_gpu_parity.py bands); what is exact here is
            # the agreement BETWEEN the layouts, checked below
            assert np.linalg.norm(r.x - o.x) <= 1e-8 * np.linalg.norm(o.x)
            sols.append(r)
    for r in sols[1:]:
        assert r.anorm == pytest.approx(sols[0].anorm, rel=1e-12)
    for k in (0, 3, 6):                             # within one layout the schedules agree bit for bit
        assert np.array_equal(sols[k].x, sols[k + 1].x) and np.array_equal(sols[k].x, sols[k + 2].x)
    assert np.array_equal(sols[0].x, sols[3].x)     # packed vs unpacked slices: the same sums


@pytest.mark.parametrize("panels", [False, True])
def test_64bit_row_pointers_path(panels):
    """nnz >= 2^31 switches the build to 64-bit row pointers (OffT = long long kernels, no sliced
    ELL).  LSQRHIP_OFF64=1 forces that path at test scale: same bits as the 32-bit build, with and
    without column panels, in all three launch schedules."""
    p = P.random_rows(30000, 40000, 9, seed=8, damp=1e-3) if panels else P.poisson2d(150, 120)
    if panels:
        os.environ.update(LSQRHIP_PANELS="1", LSQRHIP_PANEL_KB="64")
    res = []
    for off64 in ("0", "1"):
        os.environ["LSQRHIP_OFF64"] = off64
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=25)
        info = s.info()
        assert info["rowptr_bytes"] == (8 if off64 == "1" else 4)
        if off64 == "1":
            assert info["sell"] == 0
        os.environ["LSQRHIP_SELL"] = "0"          # compare like with like (row windows both times)
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=25)
        xp, yp = _vec(9, p.n), _vec(10, p.m)
        x, y = xp.copy(), yp.copy()
        s.aprod(1, p.m, p.n, x, y)
        out = [y.copy()]
        for pipeline in (0, 1, 2):
            s.set_option("pipeline", pipeline)
            r = s.solve(p.b, p.damp)
            out.append((r.x.copy(), r.anorm, r.rnorm, r.itn, r.istop))
        res.append(out)
        os.environ.pop("LSQRHIP_SELL")
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    for k in (2, 3):                                     # schedules agree with each other too
        assert np.array_equal(res[1][1][0], res[1][k][0]) and res[1][1][1:] == res[1][k][1:]
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=p.damp, itnlim=25)
    assert np.linalg.norm(res[1][1][0] - o.x) <= 1e-10 * np.linalg.norm(o.x)
