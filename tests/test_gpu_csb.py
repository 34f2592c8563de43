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
    keys = ("LSQRHIP_CSB", "LSQRHIP_CSB_R", "LSQRHIP_CSB_S", "LSQRHIP_CSB_NARROW", "LSQRHIP_CSB_XFOLD")
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
    for R, S, narrow in ((None, None, 0), (64, None, 0), (1000, None, 0), (4097, None, 0), (1000, 2, 0), (700, 3, 0),
                         (None, 5, 0), (None, None, 1), (900, 4, 1)):
        csb_env(R)
        # column splits: S workgroups share a row block and their exact sums are added by a second kernel
        os.environ.pop("LSQRHIP_CSB_S", None)
        if S:
            os.environ["LSQRHIP_CSB_S"] = str(S)
        # the 11-byte form of the index stream (u16 rows, u8 column deltas decoded by wave scans): the same bits
        os.environ["LSQRHIP_CSB_NARROW"] = str(narrow)
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=15)
        assert s.info()["col_bytes"] == (3 if narrow else 4) and s.info()["colt_bytes"] == (3 if narrow else 4)
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
    # (pipeline 0 runs the k_csb_xmax pass in front of every product, pipeline 1 takes the piece maxima from the product
    #  that wrote the vector: the same words -- test_piece_maxima_of_the_writing_product_are_those_of_the_pass)
    # ... and scaling x by a power of two scales y exactly (the grids move with max|x|)
    os.environ.pop("LSQRHIP_CSB_S", None)
    csb_env(None)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    y0, y1 = np.zeros(p.m), np.zeros(p.m)
    s.aprod(1, p.m, p.n, xp, y0)
    s.aprod(1, p.m, p.n, 2.0 ** 40 * xp, y1)
    assert np.array_equal(y1, 2.0 ** 40 * y0)


def test_the_last_split_closes_the_block_bit_for_bit_and_every_time(csb_env):
    """Round 6 (csb.h "the last split closes the block"): with column splits every split stores its exact sums write-through,
    draws a ticket of its block, and whichever split arrives LAST acquires and runs the block's epilogue from its own sums +
    the other splits' -- no k_csb_combine launch.  On the SAME row blocks (R fixed) the results are those of the unsplit
    kernel bit for bit: y of every product, and -- one partial of sum y^2 per block, the same thread -> row mapping -- the
    norms, hence every iterate of a solve.  The hand-off crosses XCDs (sums written through by one XCD's L2, read by
    another's after an agent-scope acquire): 300 products in a row on fresh vectors, each compared, would show a single stale
    line; the combine-launch form (LSQRHIP_CSB_FUSE=0) gives the same y."""
    p = P.random_rows(60000, 200000, 16, seed=21, damp=1e-3)
    os.environ.pop("LSQRHIP_CSB_FUSE", None)
    made = {}
    for S, fuse in ((1, None), (4, "1"), (8, "1"), (2, None), (4, "0")):     # (None: the build's own choice -- fused for S = 2)
        csb_env(938)
        os.environ["LSQRHIP_CSB_S"] = str(S)
        if fuse is not None:
            os.environ["LSQRHIP_CSB_FUSE"] = fuse
        try:
            s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=40)
        finally:
            os.environ.pop("LSQRHIP_CSB_FUSE", None)
            os.environ.pop("LSQRHIP_CSB_S", None)
        assert s.info()["xlds"] == 3 and s.get_option("csb_splits_mode1") == S and s.get_option("csb_blocks_mode1") == 64
        assert s.get_option("csb_fuse_mode1") == (0 if S == 1 else (1 if fuse == "1" or (fuse is None and S == 2) else 0))
        # rounds of 256 units (64 blocks x S splits), and no launch behind them where the splits close their blocks themselves
        assert s.get_option("launches_mode1") == (64 * S + 255) // 256 + (1 if fuse == "0" else 0)
        made[(S, fuse)] = s
    rs = np.random.RandomState(5)
    for k in range(300):
        x = rs.uniform(-1, 1, size=p.n) * 10.0 ** rs.randint(-3, 4)
        y0 = rs.uniform(-1, 1, size=p.m)
        outs = []
        for key, s in made.items():
            y = y0.copy()
            s.aprod(1, p.m, p.n, x.copy(), y)
            outs.append(y)
        for y in outs[1:]:
            assert np.array_equal(y, outs[0]), k
    ref = made[(1, None)].solve(p.b, p.damp)
    for key in ((4, "1"), (8, "1"), (2, None)):
        r = made[key].solve(p.b, p.damp)
        assert (r.istop, r.itn, r.anorm, r.rnorm, r.xnorm) == (ref.istop, ref.itn, ref.anorm, ref.rnorm, ref.xnorm), key
        assert np.array_equal(r.x, ref.x), key
    r = made[(4, "0")].solve(p.b, p.damp)       # (the combine launch leaves Q partials per block: the norms agree to rounding)
    assert r.itn == ref.itn and np.linalg.norm(r.x - ref.x) <= 1e-12 * np.linalg.norm(ref.x)


def test_coefficients_and_grids_handed_from_launch_to_launch_change_no_bit(csb_env):
    """Round 6 (csb.h CsbHand): the first launch of a column-swept product derives the coefficients and the two grids and
    leaves them for the product's later launches -- further rounds of row blocks -- and for its combine launch, which used
    to derive them again from the same partials and piece maxima.  Same numbers, so every bit of every product and solve
    is that of LSQRHIP_CSB_HAND=0 (every launch for itself, as in rounds 2-5): rounds of 256 units with one, two (closed by
    the last arriver) and three (combine launch) splits per block."""
    p = P.random_rows(90000, 40000, 14, seed=33, damp=1e-3)
    for S in (1, 2, 3):
        outs = []
        for hand in ("1", "0"):
            csb_env(300)                      # 300 row blocks (mode 2: 134): several rounds of 256 units at every S
            os.environ["LSQRHIP_CSB_S"] = str(S)
            os.environ["LSQRHIP_CSB_HAND"] = hand
            try:
                s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=30)
            finally:
                os.environ.pop("LSQRHIP_CSB_HAND", None)
                os.environ.pop("LSQRHIP_CSB_S", None)
            assert s.info()["xlds"] == 3 and s.get_option("launches_mode1") >= 2
            xp, yp = vecs(p)
            y = yp.copy(); s.aprod(1, p.m, p.n, xp.copy(), y)
            x = xp.copy(); s.aprod(2, p.m, p.n, x, yp.copy())
            r = s.solve(p.b, p.damp)
            outs.append((y, x, r))
        (y1, x1, r1), (y0, x0, r0) = outs
        assert np.array_equal(y1, y0) and np.array_equal(x1, x0), S
        assert (r1.istop, r1.itn, r1.anorm, r1.rnorm, r1.xnorm) == (r0.istop, r0.itn, r0.anorm, r0.rnorm, r0.xnorm), S
        assert np.array_equal(r1.x, r0.x), S


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


# ---------------------------------------------------------------------------------------------
# badly scaled systems: the grid of the exact sums is each ROW's own (csb.h, round 4)
# ---------------------------------------------------------------------------------------------
def _decades(seed, count, lo=-8.0, hi=8.0):
    """10 ** U(lo, hi), from the library's own hash (the same numbers on every box)."""
    u = 0.5 * (P.u64_to_unit(P.rng_u64(seed, 11, np.arange(count, dtype=np.uint64))) + 1.0)
    return 10.0 ** (lo + (hi - lo) * u)


def _scaled(p, what, seed=3):
    a = p.a.copy()
    if what in ("rows", "both"):
        a *= _decades(seed, p.m)[p.irow - 1]
    if what in ("columns", "both"):
        a *= _decades(seed + 1, p.n)[p.icol - 1]
    return P.Problem(p.name + "_scaled_" + what, p.m, p.n, p.irow, p.icol, a, p.b.copy(), p.damp)


def _per_row_error(mode, p, got, ref, xv):
    """max_i |got_i - ref_i| / sum_j |a_ij x_j| over the rows of the product (mode 2: of A')."""
    rows = (p.irow if mode == 1 else p.icol) - 1
    cols = (p.icol if mode == 1 else p.irow) - 1
    scale = np.zeros(p.m if mode == 1 else p.n)
    np.add.at(scale, rows, np.abs(p.a * xv[cols]))
    nz = scale > 0
    assert np.all(got[~nz] == ref[~nz])
    return float(np.max(np.abs(got[nz] - ref[nz]) / scale[nz]))


@pytest.mark.parametrize("what", ["rows", "columns", "both"])
@pytest.mark.parametrize("R,S", [(None, None), (333, None), (1000, 3)])
def test_badly_scaled_rows_and_columns_keep_every_row_accurate(csb_env, what, R, S):
    """Rows and / or columns scaled by 10^U(-8, 8) -- weighted least squares, columns in different units.  r03's
    single grid for the whole matrix left a row 2^s below the largest row 1-norm only 61 - s bits; now every
    row is held to 1e-12 of ITS OWN sum_j |a_ij x_j| against the oracle, in both modes."""
    csb_env(R)
    os.environ.pop("LSQRHIP_CSB_S", None)
    if S:
        os.environ["LSQRHIP_CSB_S"] = str(S)
    p = _scaled(P.random_rows(6000, 2500, 12, seed=17), what)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    info = s.info()
    assert info["xlds"] == 3 and info["xlds_t"] == 3
    po = oracle.port()
    xp, yp = vecs(p)
    x, y = xp.copy(), np.zeros(p.m)
    s.aprod(1, p.m, p.n, x, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, np.zeros(p.m))
    assert _per_row_error(1, p, y, y_ref, xp) <= 1e-12
    x, y = np.zeros(p.n), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, np.zeros(p.n), yp)
    assert _per_row_error(2, p, x, x_ref, yp) <= 1e-12


@pytest.mark.parametrize("spikes", [1, 3, 40])
@pytest.mark.parametrize("R,S", [(None, None), (500, 2)])
def test_a_spike_in_x_does_not_cost_the_other_rows_their_bits(csb_env, spikes, R, S):
    """x with a few entries 1e12 above the rest (v or u of a system with one badly scaled column / row): the rows that
    never touch a spike keep a grid of their own magnitude -- the big columns are summed apart, on a coarse grid,
    as integers again (csb.h "tau") -- and the rows that do touch one come out right as well."""
    csb_env(R)
    os.environ.pop("LSQRHIP_CSB_S", None)
    if S:
        os.environ["LSQRHIP_CSB_S"] = str(S)
    p = P.random_rows(20000, 9000, 10, seed=23)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    assert s.info()["xlds"] == 3 and s.info()["xlds_t"] == 3
    po = oracle.port()
    xp, yp = vecs(p)
    at = P.u64_to_index(P.rng_u64(5, 12, np.arange(spikes, dtype=np.uint64)), p.n).astype(np.int64)
    xq = xp.copy()
    xq[at] *= 1.0e12
    ys = []
    for _ in range(2):
        x, y = xq.copy(), np.zeros(p.m)
        s.aprod(1, p.m, p.n, x, y)
        ys.append(y)
    assert np.array_equal(ys[0], ys[1])            # the coarse sums are integer sums: repeatable, and cleared
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xq, np.zeros(p.m))
    assert _per_row_error(1, p, ys[0], y_ref, xq) <= 1e-12
    at = P.u64_to_index(P.rng_u64(6, 12, np.arange(spikes, dtype=np.uint64)), p.m).astype(np.int64)
    yq = yp.copy()
    yq[at] *= 1.0e12
    x, y = np.zeros(p.n), yq.copy()
    s.aprod(2, p.m, p.n, x, y)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, np.zeros(p.n), yq)
    assert _per_row_error(2, p, x, x_ref, yq) <= 1e-12
    # ... and the same matrix without spikes right after: nothing is left behind in the coarse sums
    x, y = xp.copy(), np.zeros(p.m)
    s.aprod(1, p.m, p.n, x, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, np.zeros(p.m))
    assert _per_row_error(1, p, y, y_ref, xp) <= 1e-12


@pytest.mark.parametrize("what", ["rows", "columns", "both"])
def test_short_solve_of_a_badly_scaled_system(csb_env, what):
    """10 iterations on the scaled systems against the oracle: 1e-10, or ten times what the REFERENCE's own x moves by
    when its COO input is permuted (the only freedom a row sum has: its order)."""
    csb_env(None)
    p = _scaled(P.random_rows(6000, 2500, 12, seed=17, damp=0.0), what, seed=9)
    po = oracle.port()
    kw = dict(itnlim=10)
    g = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **kw)
    q = P.shuffled(p)
    g2 = po.solve(q.m, q.n, q.irow, q.icol, q.a, q.b, **kw)
    nx = np.linalg.norm(g.x)
    band = np.linalg.norm(g2.x - g.x) / nx
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, **kw)
    assert s.info()["xlds"] == 3 and s.info()["xlds_t"] == 3
    r = s.solve(p.b, 0.0)
    err = np.linalg.norm(r.x - g.x) / nx
    print(f"scaled {what}: gpu {err:.2e}, reference under permutation {band:.2e}")
    assert (r.istop, r.itn) == (g.istop, g.itn)
    assert err <= max(1e-10, 10.0 * band)
    assert abs(r.anorm - g.anorm) <= max(1e-10, 10.0 * abs(g2.anorm - g.anorm) / g.anorm) * g.anorm


def test_values_that_do_not_survive_the_row_scaling_decline_the_layout(csb_env):
    """A value that is not finite, or one 2^-1000 below its row's largest: the stored a_ij 2^-e1_i would not be
    exact -- such a matrix keeps another layout (and the reference's floating-point row sums)."""
    csb_env(None)
    p = P.random_rows(3000, 1200, 8, seed=31)
    po = oracle.port()
    xp, yp = vecs(p)
    for bad in (np.inf, np.nan, 1e-320):
        a = p.a.copy()
        a[17] = bad
        if bad == 1e-320:
            a[p.irow == p.irow[17]] *= 1e300      # the row's other entries: 2^2000 above it
            a[17] = bad
        s = lsqr_solver_ez().initialize(p.m, p.n, a, p.irow, p.icol)
        assert s.info()["xlds"] != 3
        x, y = xp.copy(), yp.copy()
        s.aprod(1, p.m, p.n, x, y)
        with np.errstate(all="ignore"):
            _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, a, xp, yp)
        fin = np.isfinite(y_ref)
        assert np.array_equal(np.isnan(y), np.isnan(y_ref))
        assert np.max(np.abs(y[fin] - y_ref[fin]) / np.maximum(np.abs(y_ref[fin]), 1.0)) <= 1e-13


@pytest.mark.parametrize("S", [None, 2, 3])
def test_solves_with_spiky_vectors_repeat_exactly(csb_env, S):
    """A few rows and columns 1e6 above the rest: u and v carry spikes in every iteration, so every product uses the
    coarse sums in HBM.  They are integer sums and must be back at zero after every product -- also after the
    products a stopped solve leaves half done (speculative launches behind the stop flag): solves repeat bit for
    bit, across the launch schedules too."""
    csb_env(700)
    os.environ.pop("LSQRHIP_CSB_S", None)
    if S:
        os.environ["LSQRHIP_CSB_S"] = str(S)
    p = P.random_rows(9000, 4000, 9, seed=41, damp=1e-3)
    a = p.a.copy()
    a[np.isin(p.irow, [5, 1234, 8000])] *= 1.0e6
    a[np.isin(p.icol, [77, 3000])] *= 1.0e6
    s = lsqr_solver_ez().initialize(p.m, p.n, a, p.irow, p.icol, itnlim=7)
    assert s.info()["xlds"] == 3 and s.info()["xlds_t"] == 3
    g = oracle.port().solve(p.m, p.n, p.irow, p.icol, a, p.b, damp=1e-3, itnlim=7)
    runs = []
    for pipeline in (2, 2, 1, 0, 2):
        s.set_option("pipeline", pipeline)
        r = s.solve(p.b, 1e-3)
        runs.append((r.istop, r.itn, r.anorm, r.rnorm, r.xnorm, r.x.copy()))
    for o in runs[1:]:
        assert o[:5] == runs[0][:5] and np.array_equal(o[5], runs[0][5])
    assert (runs[0][0], runs[0][1]) == (g.istop, g.itn)
    assert np.linalg.norm(runs[0][5] - g.x) <= 1e-10 * np.linalg.norm(g.x)


@pytest.mark.parametrize("world,parts", [(2, 2), (3, 2), (8, 2), (8, 4), (5, 3)])
def test_overlap_plan_of_the_sharded_engine_changes_no_bit_of_a_product(world, parts):
    """LSQRHIP_SHARD_OVERLAP=1 builds a rank's layouts for exchanges in parts (csb.h "Column stripes / phases"): the
    chunks of A's row blocks are formed per part of the gathered vector and swept part by part, the row blocks of A'
    are cut per part of the output vector and launched part-major.  Row sums are exact integer sums: neither may
    change a bit of y -- launched whole (here) or phase by phase (the engine, tests/test_gpu_engine.py)."""
    keys = ("LSQRHIP_CSB", "LSQRHIP_SHARD_OVERLAP", "LSQRHIP_SHARD_WORLD", "LSQRHIP_SHARD_PARTS")
    old = {k: os.environ.get(k) for k in keys}
    # (12000 columns: A' has few enough rows for whole blocks without column splits, which the part-major launch
    # order needs -- as the 10M-row A' of a rank of config 4 has for the opposite reason)
    p = P.random_rows(60000, 12000, 10, seed=29, damp=1e-3)
    xp, yp = vecs(p)
    try:
        os.environ["LSQRHIP_CSB"] = "1"
        res = []
        for overlap in (0, 1):
            os.environ["LSQRHIP_SHARD_OVERLAP"] = str(overlap)
            os.environ["LSQRHIP_SHARD_WORLD"] = str(world)
            os.environ["LSQRHIP_SHARD_PARTS"] = str(parts)
            s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=12)
            assert s.info()["xlds"] == 3 and s.info()["xlds_t"] == 3
            ph = (s.get_option("csb_phases_mode1"), s.get_option("csb_phases_mode2"))
            assert ph == ((min(parts, 4), min(parts, 4)) if overlap else (1, 1))
            x, y = xp.copy(), yp.copy()
            s.aprod(1, p.m, p.n, x, y)
            x2, y2 = xp.copy(), yp.copy()
            s.aprod(2, p.m, p.n, x2, y2)
            r = s.solve(p.b, 1e-3)
            res.append((y, x2, r))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    # (a solve: the partial sums of |A'u|^2 are per row block, and the blocks of A' are cut differently -- rounding)
    assert (res[0][2].istop, res[0][2].itn) == (res[1][2].istop, res[1][2].itn)
    assert np.linalg.norm(res[0][2].x - res[1][2].x) <= 1e-12 * np.linalg.norm(res[0][2].x)
    _, y_ref = oracle.port().aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(res[1][0] - y_ref)) <= 1e-13 * np.max(np.abs(y_ref))


@pytest.mark.parametrize("real32", [False, True])
def test_piece_maxima_of_the_writing_product_are_those_of_the_pass(csb_env, real32):
    """Inside the loop no k_csb_xmax pass runs: the epilogue of the product that WRITES u (v) raises the piece maxima the
    pass would have left -- same pieces, same words (csb.h csb_group_max) -- and the other product takes its grids from
    them.  LSQRHIP_CSB_XFOLD=0 brings the passes back; on vectors whose grids depend on the pieces (power-law rows: spikes
    in u) a solve must not move by a bit, whatever the blocking (row blocks that start anywhere inside a group of 64
    rows, column splits whose second kernel writes y)."""
    p = P.powerlaw_rows(5000, 3000, seed=11, dmin=3, dmax=2500, damp=1e-3)
    for R, S in ((None, None), (777, None), (1000, 3), (64, None)):
        csb_env(R)
        os.environ.pop("LSQRHIP_CSB_S", None)
        if S:
            os.environ["LSQRHIP_CSB_S"] = str(S)
        out = []
        for fold in (1, 0):
            os.environ["LSQRHIP_CSB_XFOLD"] = str(fold)
            s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=40, real32=real32)
            assert s.info()["xlds"] == 3 and s.info()["xlds_t"] == 3
            r = s.solve(p.b, 1e-3)
            out.append((r.x.copy(), r.anorm, r.rnorm, r.arnorm, r.itn))
            r = s.solve(p.b, 1e-3)    # (the second solve of a handle starts from sets the first one left)
            assert np.array_equal(r.x, out[-1][0]) and r.anorm == out[-1][1]
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:]


@pytest.mark.parametrize("shape", ["rows of 9", "powerlaw rows of 2500", "long sweeps", "overlap plan", "real32"])
def test_lock_step_sweep_changes_no_bit(csb_env, shape):
    """Round 5: the 16 waves of a workgroup move in lock step -- gathers of a step, ONE barrier, then the next step's
    stream -- K = 1 or 2 chunks per wave and step (csb.h "LOCK STEP"; LSQRHIP_CSB_LOCKSTEP, by shape when unset), where
    rounds 2-4 let every wave run on its own (LSQRHIP_CSB_LOCKSTEP=0).  A schedule, not arithmetic: row sums are exact
    integer sums, so products and solves must be IDENTICAL to the last bit in all three forms -- over blockings with
    fewer chunks than waves, odd and even step counts, column splits, the 11-byte stream, the stripes of the sharded
    engine's overlap plan (several chunk ranges per unit), REAL32 storage, and with x that holds inf / NaN / huge entries
    (the outlier pass reads the stream a second time)."""
    extra = {}
    real32 = False
    if shape == "rows of 9":
        p = P.random_rows(6000, 2500, 9, seed=7, damp=1e-3)
        blockings = ((None, None, 0), (64, None, 0), (1000, 2, 0), (700, 3, 1), (4097, None, 1))
    elif shape == "powerlaw rows of 2500":
        p = P.powerlaw_rows(5000, 3000, seed=7, dmin=3, dmax=2500, damp=1e-3)
        blockings = ((None, None, 0), (333, None, 0), (900, 4, 1))
    elif shape == "long sweeps":       # hundreds of chunks per wave and unit: many steps, both parities of the step count
        p = P.random_rows(40000, 3000, 60, seed=11, damp=1e-3)
        blockings = ((None, None, 0), (20000, None, 0), (13000, None, 1), (None, 2, 0))
    elif shape == "overlap plan":
        p = P.random_rows(60000, 12000, 10, seed=29, damp=1e-3)
        blockings = ((None, None, 0),)
        extra = {"LSQRHIP_SHARD_OVERLAP": "1", "LSQRHIP_SHARD_WORLD": "3", "LSQRHIP_SHARD_PARTS": "3"}
    else:
        p = P.random_rows(6000, 2500, 9, seed=7, damp=1e-3)
        blockings = ((None, None, 0), (1000, 2, 0))
        real32 = True
    wp = np.float32 if real32 else np.float64
    xp, yp = vecs(p)
    xbad = xp.copy()
    xbad[[3, 77, 1500]] = [np.inf, np.nan, 1e300]
    if real32:
        xbad[1500] = 3e38
    a = p.a.astype(wp)
    keys = ("LSQRHIP_CSB_LOCKSTEP",) + tuple(extra)
    old = {k: os.environ.get(k) for k in keys}
    try:
        os.environ.update(extra)
        for R, S, narrow in blockings:
            csb_env(R)
            os.environ.pop("LSQRHIP_CSB_S", None)
            if S:
                os.environ["LSQRHIP_CSB_S"] = str(S)
            os.environ["LSQRHIP_CSB_NARROW"] = str(narrow)
            out = []
            for ls in ("0", "1", "2"):
                os.environ["LSQRHIP_CSB_LOCKSTEP"] = ls
                s = lsqr_solver_ez().initialize(p.m, p.n, a, p.irow, p.icol, itnlim=12, real32=real32)
                assert s.info()["xlds"] == 3 and s.info()["xlds_t"] == 3
                assert s.get_option("csb_lockstep_mode1") == int(ls) and s.get_option("csb_lockstep_mode2") == int(ls)
                res = []
                for xin in (xp, xbad):
                    x, y = xin.astype(wp), yp.astype(wp)
                    s.aprod(1, p.m, p.n, x, y)
                    res.append(y)
                x, y = xp.astype(wp), yp.astype(wp)
                s.aprod(2, p.m, p.n, x, y)
                res.append(x)
                r = s.solve(p.b.astype(wp), 1e-3)
                res += [r.x, np.array([r.itn, r.istop]), np.array([r.anorm, r.rnorm, r.arnorm, r.xnorm])]
                out.append(res)
            for other in out[1:]:
                for u, v in zip(out[0], other):
                    assert np.array_equal(u, v, equal_nan=True), (shape, R, S, narrow)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_lock_step_by_shape_default():
    """Unset, the build chooses the chunks per step by shape: 1, or 2 for long sweeps over sparse columns."""
    old = os.environ.pop("LSQRHIP_CSB_LOCKSTEP", None)
    try:
        os.environ["LSQRHIP_CSB"] = "1"
        p = P.random_rows(6000, 2500, 9, seed=7, damp=1e-3)
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
        assert s.get_option("csb_lockstep_mode1") == 1 and s.get_option("csb_lockstep_mode2") == 1
    finally:
        os.environ.pop("LSQRHIP_CSB", None)
        if old is not None:
            os.environ["LSQRHIP_CSB_LOCKSTEP"] = old
