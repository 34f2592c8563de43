"""Column-swept row blocks (csrc/csb.h): the layout of scattered matrices whose x exceeds L2.

Forced on at test scale (LSQRHIP_CSB=1) and checked against the oracle for both aprod modes and
for solves; the exact integer sums make every result independent of the blocking, so products
under different block sizes must agree BIT FOR BIT; a matrix whose blocks are too empty for 18-bit
local columns must fall back to another layout; non-finite x must come out as the reference's."""
import os

import numpy as np
import pytest

import oracle
from cases import build_cases
from lsqr_amd import problems as P
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu
CASES = build_cases()


@pytest.fixture
def csb_env():
    keys = ("LSQRHIP_CSB", "LSQRHIP_CSB_R", "LSQRHIP_CSB_S")
    old = {k: os.environ.get(k) for k in keys}
    os.environ["LSQRHIP_CSB"] = "1"

    def set_r(r):
        if r is None:
            os.environ.pop("LSQRHIP_CSB_R", None)
        else:
            os.environ["LSQRHIP_CSB_R"] = str(r)
    yield set_r
    for k, v in old.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def vecs(p):
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    return xp, yp


@pytest.mark.parametrize("name", ["random_over_damped", "random_under", "shuffled_dups", "powerlaw_small",
                                  "empty_rows_cols", "poisson_20x20_it50", "itnlim_1", "t1_readme_damped",
                                  "one_by_one", "zero_matrix", "b_zero"])
@pytest.mark.parametrize("R", [None, 64, 333])
def test_products_and_solve_match_oracle(csb_env, name, R):
    csb_env(R)
    p, o = CASES[name]
    # the layout check; long runs are the golden tests' business.  (Power-law rows of 700: the reference's
    # own x moves by 1e-9 / 2.5e-5 after 20 / 30 iterations when its COO input is permuted)
    itn = min(o["itnlim"], 8 if name == "powerlaw_small" else 25)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=o["atol"], btol=o["btol"],
                                    conlim=o["conlim"], itnlim=itn)
    info = s.info()
    assert info["xlds"] == 3 and info["xlds_t"] == 3
    po = oracle.port()
    xp, yp = vecs(p)
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.array_equal(x, xp)
    assert np.max(np.abs(y - y_ref)) <= 1e-13 * max(np.max(np.abs(y_ref)), 1.0)
    x, y = xp.copy(), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.array_equal(y, yp)
    assert np.max(np.abs(x - x_ref)) <= 1e-13 * max(np.max(np.abs(x_ref)), 1.0)
    r = s.solve(p.b, o["damp"])
    g = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=o["damp"], atol=o["atol"], btol=o["btol"],
                 conlim=o["conlim"], itnlim=itn)
    assert r.istop == g.istop
    nx = np.linalg.norm(g.x)
    if nx > 0:
        assert np.linalg.norm(r.x - g.x) <= 1e-10 * nx
    else:
        assert np.all(r.x == 0.0)
    if name != "t1_readme_damped":      # (stops at the eps level after 4 iterations)
        assert r.itn == g.itn
    if g.itn > 0 and r.itn == g.itn:
        assert abs(r.anorm - g.anorm) <= 1e-10 * g.anorm
        assert abs(r.rnorm - g.rnorm) <= 1e-10 * g.rnorm or g.rnorm <= 1e-13 * np.linalg.norm(p.b)


@pytest.mark.parametrize("kind", ["powerlaw rows of 2500", "rows of 9"])
def test_results_do_not_depend_on_the_blocking_bit_for_bit(csb_env, kind):
    """Every product is rounded once onto a fixed binary grid and the sums on that grid are 64-bit integer
    sums: exact, so the order of the adds -- and with it the block size, the column splits, the launch shape,
    which wave took which chunk -- cannot change a bit of y.  A solve inherits that up to its partial sums of y^2
    (one per block, reduced in block order): identical blockings repeat exactly."""
    if kind.startswith("powerlaw"):
        p = P.powerlaw_rows(5000, 3000, seed=7, dmin=3, dmax=2500, damp=1e-3)
    else:
        p = P.random_rows(6000, 2500, 9, seed=7, damp=1e-3)      # A: 9 per row; A': ~22 per row
    xp, yp = vecs(p)
    ys, xs = [], []
    for R, S in ((None, None), (64, None), (1000, None), (4097, None), (1000, 2), (700, 3), (None, 5)):
        csb_env(R)
        # column splits: S workgroups share a row block and their exact sums are added by a second kernel
        os.environ.pop("LSQRHIP_CSB_S", None)
        if S:
            os.environ["LSQRHIP_CSB_S"] = str(S)
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=15)
        x, y = xp.copy(), yp.copy()
        s.aprod(1, p.m, p.n, x, y)
        ys.append(y)
        x, y = xp.copy(), yp.copy()
        s.aprod(2, p.m, p.n, x, y)
        xs.append(x)
        r1, r2 = s.solve(p.b, 1e-3), s.solve(p.b, 1e-3)
        assert np.array_equal(r1.x, r2.x) and (r1.anorm, r1.rnorm, r1.itn) == (r2.anorm, r2.rnorm, r2.itn)
        for pipeline in (0, 1):
            s.set_option("pipeline", pipeline)
            r3 = s.solve(p.b, 1e-3)
            assert np.array_equal(r1.x, r3.x) and r1.anorm == r3.anorm
    for y in ys[1:]:
        assert np.array_equal(y, ys[0])
    for x in xs[1:]:
        assert np.array_equal(x, xs[0])
    # ... and scaling x by a power of two scales y exactly (the grids move with max|x|)
    os.environ.pop("LSQRHIP_CSB_S", None)
    csb_env(None)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    y0, y1 = np.zeros(p.m), np.zeros(p.m)
    s.aprod(1, p.m, p.n, xp, y0)
    s.aprod(1, p.m, p.n, 2.0 ** 40 * xp, y1)
    assert np.array_equal(y1, 2.0 ** 40 * y0)


def test_a_block_too_empty_for_local_columns_falls_back(csb_env):
    """18-bit local columns need every chunk of 256 column-sorted nonzeros to span < 262144 columns.
    Three nonzeros per row over 600000 columns in blocks of 64 rows do not: the build must notice and
    use another layout, with correct results."""
    csb_env(64)
    p = P.random_rows(2000, 600000, 3, seed=9)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=5)
    info = s.info()
    assert info["xlds"] != 3          # mode 1 gathers over 600000 columns: too wide
    po = oracle.port()
    xp, yp = vecs(p)
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(y - y_ref)) <= 1e-13 * max(np.max(np.abs(y_ref)), 1.0)
    x, y = xp.copy(), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(x - x_ref)) <= 1e-13 * max(np.max(np.abs(x_ref)), 1.0)


def test_non_finite_and_huge_x_as_the_reference(csb_env):
    csb_env(128)
    p = P.random_rows(700, 300, 6, seed=21)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    po = oracle.port()
    xp, yp = vecs(p)
    for bad in (np.inf, -np.inf, np.nan, 1e300, 1e-300):
        xq = xp.copy()
        xq[17] = bad
        xq[250] = bad if bad != 1e-300 else 1e-310
        x, y = xq.copy(), yp.copy()
        with np.errstate(all="ignore"):
            _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xq, yp)
        s.aprod(1, p.m, p.n, x, y)
        fin = np.isfinite(y_ref)
        assert np.array_equal(np.isnan(y), np.isnan(y_ref))
        assert np.array_equal(y[~fin & ~np.isnan(y_ref)], y_ref[~fin & ~np.isnan(y_ref)])     # the same infinities
        assert np.max(np.abs(y[fin] - y_ref[fin])) <= 1e-13 * max(np.max(np.abs(y_ref[fin])), 1.0)


@pytest.mark.parametrize("shape", [(30000, 400000, 6), (200000, 300000, 12)])
def test_default_choice_at_moderate_scale(shape):
    """Without any knob: scattered columns over an x beyond two L2 panels select the layout by
    themselves (A gathers over n columns; A' may or may not qualify); oracle parity of a short solve."""
    m, n, per = shape
    p = P.random_rows(m, n, per, seed=5, damp=1e-3)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=10)
    info = s.info()
    assert info["xlds"] == 3 or n <= 524288
    r = s.solve(p.b, 1e-3)
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=1e-3, itnlim=10)
    assert (r.istop, r.itn) == (o.istop, o.itn)
    assert np.linalg.norm(r.x - o.x) <= 1e-10 * np.linalg.norm(o.x)
    assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * o.rnorm
