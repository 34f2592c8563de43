"""Row patterns (csrc/pat.h): a matrix whose rows repeat -- a constant-coefficient stencil -- is stored as one byte per
row plus the table of its distinct rows.  Like every other short-row layout it must give the reference's row sums bit
for bit (src/lsqr.f90:166-174, 186-194: left to right in COO order) and the reference's solve; a matrix that does
not qualify, or one whose rows collide under the hash, keeps the layout it would have had."""
import os

import numpy as np
import pytest

import oracle
from cases import build_cases
from lsqr_amd import problems as P
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu
CASES = build_cases()
KNOBS = ("LSQRHIP_STREAM_NT", "LSQRHIP_PAT", "LSQRHIP_PAT2", "LSQRHIP_PAT_PAIR", "LSQRHIP_SPAT", "LSQRHIP_SELL", "LSQRHIP_SELLP", "LSQRHIP_VAL8", "LSQRHIP_COL16", "LSQRHIP_CSB")


@pytest.fixture(autouse=True)
def clean_env():
    old = {k: os.environ.pop(k, None) for k in KNOBS}
    yield
    for k, v in old.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def _vec(stream, n):
    return P.u64_to_unit(P.rng_u64(301, stream, np.arange(n, dtype=np.uint64)))


def stencil(m, n, offsets, values, seed=1, shuffle=False, drop=None):
    """Row r holds values[k] at column r + offsets[k] where that is inside [0, n), in the order given (duplicates and
    unsorted offsets allowed: the order IS the summation order)."""
    r = np.arange(m)
    rows, cols, vals = [], [], []
    for k, (off, v) in enumerate(zip(offsets, values)):
        c = r + off
        ok = (c >= 0) & (c < n)
        if drop is not None:
            ok &= ~drop(r, k)
        rows.append(np.where(ok, r, -1)); cols.append(c); vals.append(np.full(m, v))
    rows = np.stack(rows, axis=1).ravel(); cols = np.stack(cols, axis=1).ravel(); vals = np.stack(vals, axis=1).ravel()
    keep = rows >= 0
    irow, icol, a = rows[keep], cols[keep], vals[keep]
    if shuffle:      # entry k of every row, then entry k - 1 of every row, ..., each time from the last row to the first:
        # a stable sort by row leaves every row with its entries reversed
        kk = np.stack([np.full(m, k) for k in range(len(offsets))], axis=1).ravel()[keep]
        perm = np.lexsort((-irow, -kk))
        irow, icol, a = irow[perm], icol[perm], a[perm]
    b = _vec(7, m)
    return m, n, (irow + 1).astype(np.int32), (icol + 1).astype(np.int32), a.astype(np.float64), b


def check_against_oracle(m, n, irow, icol, a, b, damp=0.0, itnlim=25, want=3, want_t=3):
    po = oracle.port()
    xp, yp = _vec(9, n), _vec(10, m)
    _, y_ref = po.aprod(1, m, n, irow, icol, a, xp, yp)
    x_ref, _ = po.aprod(2, m, n, irow, icol, a, xp, yp)
    o = po.solve(m, n, irow, icol, a, b, damp=damp, itnlim=itnlim)
    s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=itnlim)
    info = s.info()
    assert (info["sell"], info["sell_t"]) == (want, want_t), info
    x, y = xp.copy(), yp.copy()
    s.aprod(1, m, n, x, y)
    assert np.array_equal(y, y_ref)               # the reference's row sums, bit for bit
    x, y = xp.copy(), yp.copy()
    s.aprod(2, m, n, x, y)
    assert np.array_equal(x, x_ref)
    outs = []
    for pipeline in (0, 1, 2):
        s.set_option("pipeline", pipeline)
        r = s.solve(b, damp)
        assert (r.istop, r.itn) == (o.istop, o.itn)
        assert np.linalg.norm(r.x - o.x) <= 1e-10 * max(np.linalg.norm(o.x), 1e-300)
        assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * max(o.rnorm, 1e-300)
        outs.append(r)
    for r in outs[1:]:
        assert np.array_equal(r.x, outs[0].x) and r.anorm == outs[0].anorm and r.rnorm == outs[0].rnorm
    return s, outs[0]


@pytest.mark.parametrize("shape", [(300, 200), (64, 3), (257, 131), (1000, 2)])
def test_a_stencil_is_stored_as_row_patterns_and_gives_the_references_bits(shape):
    p = P.poisson2d(*shape)
    s, r = check_against_oracle(p.m, p.n, p.irow, p.icol, p.a, p.b)
    info = s.info()
    # a byte per row and the table: far below the 16 bytes per row of the packed records
    assert info["csr_bytes"] <= p.m + 1024 + 12 * 1024
    # ... and the layouts underneath: the same bits from the slice form of the pattern kernel (lane L owns row L of a
    # 64-row slice, as in sell.h: the same rows per thread, the same partial sums of the norms); the paired form -- the
    # default: lane L owns rows 2L, 2L + 1 -- forms the same row sums and adds their squares in another order
    os.environ["LSQRHIP_PAT_PAIR"] = "0"
    s1 = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=25)
    assert s1.info()["sell"] == 3 and s1.get_option("pat_pair_mode1") == 0 and s.get_option("pat_pair_mode1") == (1 if p.n >= 2 else 0)
    r1 = s1.solve(p.b, 0.0)
    os.environ["LSQRHIP_PAT"] = "0"
    os.environ["LSQRHIP_SPAT"] = "0"
    s0 = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=25)
    assert s0.info()["sell"] in (1, 2)
    r0 = s0.solve(p.b, 0.0)
    assert np.array_equal(r0.x, r1.x) and r0.anorm == r1.anorm and r0.rnorm == r1.rnorm and r0.itn == r1.itn
    assert r0.itn == r.itn and np.linalg.norm(r0.x - r.x) <= 1e-12 * np.linalg.norm(r.x)
    assert abs(r0.anorm - r.anorm) <= 1e-13 * r.anorm and abs(r0.rnorm - r.rnorm) <= 1e-12 * r.rnorm
    # the products themselves: bit for bit, paired or not
    xp, yp = _vec(9, p.n), _vec(10, p.m)
    outs = []
    for sv in (s, s1, s0):
        x, y = xp.copy(), yp.copy()
        sv.aprod(1, p.m, p.n, x, y)
        x2, y2 = xp.copy(), yp.copy()
        sv.aprod(2, p.m, p.n, x2, y2)
        outs.append((y, x2))
    assert all(np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) for o in outs[1:])


@pytest.mark.parametrize("m,n,offs", [
    (127, 130, (-1, 0, 1)),                          # fewer rows than one 128-row pair group, odd
    (1, 2, (0, 1)),                                  # one row
    (1001, 2, (0, -1, 1)),                           # two columns, odd rows: the last pair is half empty
    (4097, 4100, (-9, -5, -2, -1, 0, 1, 2, 5, 9)),   # 9 entries per row: longer than PAT_K (5) entries in flight
    (255, 255, (-3, -1, 0, 0, 1, 3, 7)),             # a duplicate entry, 7 long, odd
])
def test_paired_rows_and_the_slice_form_agree(m, n, offs):
    """The contract of the paired-row kernels (k_spmv_patp / k_spmv_pat2p, the default): against the slice
    form (LSQRHIP_PAT_PAIR=0) every PRODUCT is bit-equal -- the same left-to-right row sums -- while the fused norms add
    the rows' squares in another order, so alpha / beta may differ in the last bit and a solve agrees to rounding, not
    bit for bit."""
    vals = tuple(0.37 * (k + 1) * (-1) ** k for k in range(len(offs)))
    m, n, irow, icol, a, b = stencil(m, n, offs, vals)
    os.environ["LSQRHIP_PAT"] = "1"
    made = {}
    for pair in ("1", "0"):
        os.environ["LSQRHIP_PAT_PAIR"] = pair
        s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=12)
        assert s.info()["sell"] == 3 and s.info()["sell_t"] == 3
        made[pair] = s
    assert made["0"].get_option("pat_pair_mode1") == 0
    if n >= 2 and m >= 2:
        assert made["1"].get_option("pat_pair_mode1") == 1
    xp, yp = _vec(9, n), _vec(10, m)
    outs = {}
    for pair, s in made.items():
        y = yp.copy(); s.aprod(1, m, n, xp.copy(), y)
        x = xp.copy(); s.aprod(2, m, n, x, yp.copy())
        outs[pair] = (y, x, s.solve(b, 1e-2))
    assert np.array_equal(outs["0"][0], outs["1"][0]) and np.array_equal(outs["0"][1], outs["1"][1])
    r0, r1 = outs["0"][2], outs["1"][2]
    assert (r0.istop, r0.itn) == (r1.istop, r1.itn)
    assert np.linalg.norm(r0.x - r1.x) <= 1e-12 * max(np.linalg.norm(r0.x), 1e-300)
    assert abs(r0.anorm - r1.anorm) <= 1e-13 * r0.anorm and abs(r0.rnorm - r1.rnorm) <= 1e-12 * max(r0.rnorm, 1e-300)
    po = oracle.port()
    _, y_ref = po.aprod(1, m, n, irow, icol, a, xp, yp)
    assert np.array_equal(outs["1"][0], y_ref)


def test_unsymmetric_patterns_duplicates_and_the_order_inside_a_row():
    """An upwind-like stencil whose transpose has other values, one offset twice (a duplicate (i, j) entry), offsets
    out of order, rows delivered last-first: the stable sort by row keeps the order inside each row, and that order is
    the summation order of the pattern."""
    offs = (3, -2, 0, 3, 1, -40)
    vals = (0.3, -1.7, 2.0 / 3.0, 1e-3, -0.0, 5.5)
    m, n, irow, icol, a, b = stencil(5000, 5007, offs, vals, shuffle=True)
    check_against_oracle(m, n, irow, icol, a, b, damp=1e-2)


def test_rows_with_holes_make_more_patterns():
    """Every 7th row lacks its second entry, every 11th its fourth: 4 interior patterns and the boundary ones."""
    drop = lambda r, k: ((k == 1) & (r % 7 == 0)) | ((k == 3) & (r % 11 == 0))   # noqa: E731
    m, n, irow, icol, a, b = stencil(9000, 9000, (-5, -1, 0, 1, 5), (-1.0, -1.0, 4.0, -1.0, -1.0), drop=drop)
    check_against_oracle(m, n, irow, icol, a, b)


def test_empty_rows_are_a_pattern_of_their_own():
    drop = lambda r, k: r % 5 == 2          # noqa: E731
    m, n, irow, icol, a, b = stencil(4000, 3990, (-1, 0, 2), (1.0, -2.0, 1.0), drop=drop)
    check_against_oracle(m, n, irow, icol, a, b, damp=1e-3)


def test_matrices_without_repeating_rows_keep_their_layout():
    # arbitrary real values on a band: every row is its own pattern -- but the STRUCTURE repeats (sell = 4, below)
    m, n, irow, icol, a, b = stencil(70001, 70003, (-7, -1, 0, 1, 7), (1.0,) * 5)
    a = _vec(3, a.size) + 0.5
    info = lsqr_solver_ez().initialize(m, n, a, irow, icol).info()
    assert info["sell"] == 4 and info["sell_t"] == 4
    os.environ["LSQRHIP_SPAT"] = "0"
    info = lsqr_solver_ez().initialize(m, n, a, irow, icol).info()
    assert info["sell"] == 1 and info["sell_t"] == 1
    os.environ.pop("LSQRHIP_SPAT")
    # random columns
    p = P.random_rows(30000, 20000, 8, seed=4)
    info = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()
    assert info["sell"] != 3 and info["sell_t"] != 3
    # nd distinct diagonal values, each 100 times, over a constant superdiagonal that the last row lacks: nd + 1
    # patterns -- 256 fit a byte, 257 do not (they take two: the wide table below; without it the structure -- two
    # patterns -- is what is left, one value too many for a dictionary as well)
    for nd, want in ((255, 3), (256, 1)):
        m, n, irow, icol, a = diagonal_steps(nd, 100)
        if want == 3:
            s, _ = check_against_oracle(m, n, irow, icol, a, _vec(7, m), itnlim=8)
            assert s.info()["pat_wide"] == 0
        else:
            os.environ["LSQRHIP_PAT2"] = "0"
            assert lsqr_solver_ez().initialize(m, n, a, irow, icol).info()["sell"] in (want, 4)
            os.environ.pop("LSQRHIP_PAT2")


def diagonal_steps(nd, each):
    """nd distinct diagonal values, each `each` times in turn, over a constant superdiagonal that the last row lacks:
    nd + 1 distinct rows."""
    m = n = each * nd
    r = np.arange(m)
    irow = np.concatenate([r, r[:-1]])
    icol = np.concatenate([r, r[:-1] + 1])
    a = np.concatenate([1.0 + (r % nd), np.full(m - 1, -1.0)])
    order = np.argsort(irow, kind="stable")
    return m, n, (irow[order] + 1).astype(np.int32), (icol[order] + 1).astype(np.int32), a[order]


def grid3d(nx, ny, nz, radius_taps):
    """Row (i, j, k) of an nx x ny x nz grid: one entry per tap (di, dj, dk) that stays inside the grid, value a
    function of the tap only (constant coefficients), taps in a fixed order."""
    taps = [(di, dj, dk) for dk in (-1, 0, 1) for dj in (-1, 0, 1) for di in (-1, 0, 1)
            if abs(di) + abs(dj) + abs(dk) <= radius_taps]
    idx = np.arange(nx * ny * nz)
    i, j, k = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    rows, cols, vals = [], [], []
    for t, (di, dj, dk) in enumerate(taps):
        ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
        rows.append(np.where(ok, idx, -1)); cols.append(idx + di + nx * dj + nx * ny * dk)
        vals.append(np.full(idx.size, 26.0 if (di, dj, dk) == (0, 0, 0) else -1.0 / (1 + abs(di) + abs(dj) + abs(dk)) - 0.001 * t))
    rows = np.stack(rows, axis=1).ravel(); cols = np.stack(cols, axis=1).ravel(); vals = np.stack(vals, axis=1).ravel()
    keep = rows >= 0
    m = idx.size
    return m, m, (rows[keep] + 1).astype(np.int32), (cols[keep] + 1).astype(np.int32), vals[keep], _vec(7, m), len(taps)


@pytest.mark.parametrize("taps", [1, 3])
def test_three_dimensional_stencils(taps):
    """7-point (radius 1 in the 1-norm) and 27-point operators on a 3-D grid: 27 boundary variants each, 343 pattern
    entries for the 27-point one; unsymmetric values, so A' has patterns of its own."""
    m, n, irow, icol, a, b, ntaps = grid3d(24, 19, 17, taps)
    assert ntaps == (7 if taps == 1 else 27)
    check_against_oracle(m, n, irow, icol, a, b, itnlim=12)


def test_a_convolution_of_31_taps():
    """Banded Toeplitz: y = k * x with a 31-tap kernel, zero boundary -- 31 patterns, ~730 entries."""
    offs = tuple(range(-15, 16))
    vals = tuple(np.exp(-0.02 * o * o) * (1.0 + 0.1 * np.sin(o)) for o in offs)
    m, n, irow, icol, a, b = stencil(30000, 30000, offs, vals)
    check_against_oracle(m, n, irow, icol, a, b, damp=1e-3, itnlim=10)


def test_limits_rows_per_pattern_entries_and_row_length():
    # 3 x 3 dense (README example): three patterns for three rows -- only when forced
    p, o = CASES["t1_readme_damped"]
    assert lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()["sell"] != 3
    os.environ["LSQRHIP_PAT"] = "1"
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=o["itnlim"])
    assert s.info()["sell"] == 3 and s.info()["sell_t"] == 3
    r = s.solve(p.b, o["damp"])
    ref = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=o["damp"], itnlim=o["itnlim"])
    assert r.istop == ref.istop and np.linalg.norm(r.x - ref.x) <= 1e-10 * np.linalg.norm(ref.x)
    os.environ.pop("LSQRHIP_PAT")
    # 60 entries per row, interior + 2 x 30 boundary patterns: more than 1024 entries in all
    offs = tuple(range(-30, 30))
    m, n, irow, icol, a, b = stencil(20000, 20000, offs, tuple(1.0 + 0.01 * k for k in range(60)))
    assert lsqr_solver_ez().initialize(m, n, a, irow, icol).info()["sell"] != 3
    # 12 entries per row: 1 + 2 x 6 patterns, ~130 entries
    offs = tuple(range(-6, 6))
    m, n, irow, icol, a, b = stencil(20000, 20000, offs, tuple(1.0 + 0.01 * k for k in range(12)))
    check_against_oracle(m, n, irow, icol, a, b, itnlim=10)
    # rows of 65 nonzeros: beyond the row length a pattern may have
    offs = tuple(range(-32, 33))
    m, n, irow, icol, a, b = stencil(5000, 5000, offs, (1.0,) * 65)
    assert lsqr_solver_ez().initialize(m, n, a, irow, icol).info()["sell"] != 3


def test_real32_patterns():
    """REAL32 handle on the pattern layout: float vectors, binary64 registers, the table stays binary64 (of values that
    are exactly float)."""
    p = P.poisson2d(120, 90)
    a32, b32 = p.a.astype(np.float32), p.b.astype(np.float32)
    s = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, itnlim=30, real32=True)
    assert s.info()["sell"] == 3
    xp, yp = _vec(9, p.n).astype(np.float32), _vec(10, p.m).astype(np.float32)
    po = oracle.port()
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, a32.astype(np.float64), xp.astype(np.float64), yp.astype(np.float64))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    assert np.array_equal(y, y_ref.astype(np.float32))      # binary64 sums, rounded once
    os.environ["LSQRHIP_PAT"] = "0"
    s0 = lsqr_solver_ez().initialize(p.m, p.n, a32, p.irow, p.icol, itnlim=30, real32=True)
    r, r0 = s.solve(b32, 0.0), s0.solve(b32, 0.0)
    assert np.array_equal(r.x, r0.x) and r.itn == r0.itn and r.anorm == r0.anorm


@pytest.mark.parametrize("pair", ["1", "0"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_the_golden_parity_cases_with_patterns_forced(name, pair):
    """Every case of tests/test_gpu_parity.py once more with LSQRHIP_PAT=1: whatever has <= 256 distinct rows of <= 64
    nonzeros and <= 1024 entries -- most of the small systems: every row its own pattern -- goes through the pattern
    kernel (paired rows, the default, and the slice form) and must hold the same golden values; the others keep
    their layout."""
    import test_gpu_parity as tp
    os.environ["LSQRHIP_PAT"] = "1"
    os.environ["LSQRHIP_PAT_PAIR"] = pair
    tp.test_solve_parity_vs_reference_golden(name)


# ---------------------------------------------------------------------------------------------------------------------
# wide row patterns (sell = 3, two-byte pattern numbers, the table through L2): 257 ... 4096 distinct rows
# ---------------------------------------------------------------------------------------------------------------------
def piecewise_mesh(nx, ny, bx, by, seed=11):
    """-div(k grad u) on an nx x ny grid, five points, k constant on each of bx x by blocks of cells (its own value on
    every block) and the harmonic mean of the two sides on a face: one row for the interior of a block, others along
    every interface, at every corner where four blocks meet, and along the boundary of the domain
    (lsqr_amd/problems.py mesh2d, the host twin of the generator bench.py uses)."""
    p = P.mesh2d(nx, ny, bx, by, seed=seed)
    return p.m, p.n, p.irow, p.icol, p.a, p.b


def test_a_piecewise_constant_coefficient_mesh_takes_two_bytes_per_row():
    m, n, irow, icol, a, b = piecewise_mesh(500, 400, 12, 10)     # (<= 1024 blocks of 256 rows: one grid for every layout)
    s, r = check_against_oracle(m, n, irow, icol, a, b, damp=1e-3)
    info = s.info()
    assert 256 < info["pat_wide"] <= 4096 and 256 < info["pat_wide_t"] <= 4096, info
    # two bytes per row and the table (<= 5 entries of 16 bytes and a descriptor per pattern)
    assert info["csr_bytes"] <= 2 * m + 84 * info["pat_wide"]
    # ... and the layouts underneath -- without the wide table, then without any pattern layout: the same bits from the
    # slice form of the wide kernel (the same rows per thread as sell.h), the same solve to rounding from the paired form
    os.environ["LSQRHIP_PAT_PAIR"] = "0"
    s1 = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=25)
    assert s1.info()["pat_wide"] == info["pat_wide"] and s1.get_option("pat_pair_mode1") == 0 and s.get_option("pat_pair_mode1") == 1
    r1 = s1.solve(b, 1e-3)
    for knobs in ({"LSQRHIP_PAT2": "0"}, {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0"}):
        os.environ.update(knobs)
        s0 = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=25)
        assert s0.info()["pat_wide"] == 0 and s0.info()["csr_bytes"] > 2 * info["csr_bytes"]
        r0 = s0.solve(b, 1e-3)
        assert np.array_equal(r0.x, r1.x) and r0.anorm == r1.anorm and r0.rnorm == r1.rnorm and r0.itn == r1.itn
        assert r0.itn == r.itn and np.linalg.norm(r0.x - r.x) <= 1e-12 * np.linalg.norm(r.x)
        assert abs(r0.anorm - r.anorm) <= 1e-13 * r.anorm and abs(r0.rnorm - r.rnorm) <= 1e-12 * r.rnorm
        for k in knobs:
            os.environ.pop(k)
    # the products themselves: bit for bit, paired or not (check_against_oracle held s to the oracle's)
    xp, yp = _vec(9, n), _vec(10, m)
    ya, yb_ = yp.copy(), yp.copy()
    s.aprod(1, m, n, xp.copy(), ya)
    s1.aprod(1, m, n, xp.copy(), yb_)
    assert np.array_equal(ya, yb_)


@pytest.mark.parametrize("nd,wide", [(256, True), (4095, True), (4096, False)])
def test_limits_of_the_wide_table(nd, wide):
    """nd + 1 distinct rows: 257 are the first that take two bytes, 4096 the last that fit."""
    m, n, irow, icol, a = diagonal_steps(nd, 100 if nd == 256 else 20)
    if wide:
        s, _ = check_against_oracle(m, n, irow, icol, a, _vec(7, m), itnlim=8)
        assert s.info()["pat_wide"] == nd + 1 and s.info()["pat_wide_t"] == nd + 1
    else:
        info = lsqr_solver_ez().initialize(m, n, a, irow, icol).info()
        assert info["pat_wide"] == 0 and info["sell"] != 3


def test_wide_patterns_with_long_rows_and_few_rows_per_pattern():
    """12 entries per row (more than one trip of 5 through the table), every entry scaled by the coefficient of the row's
    region: a pattern per region and its boundary variants; then a matrix with fewer than 16 rows per pattern -- the
    wide table only when forced."""
    offs = tuple(range(-6, 6))
    m, n, irow, icol, a, b = stencil(90000, 90000, offs, tuple(1.0 + 0.01 * k for k in range(12)))
    a = a * (1.0 + ((irow - 1) // 300))          # 300 regions of 300 rows
    s, _ = check_against_oracle(m, n, irow, icol, a, b, itnlim=10)
    assert s.info()["pat_wide"] >= 300
    m, n, irow, icol, a = diagonal_steps(600, 10)       # 601 patterns for 6000 rows
    assert lsqr_solver_ez().initialize(m, n, a, irow, icol).info()["pat_wide"] == 0
    os.environ["LSQRHIP_PAT2"] = "1"
    s, _ = check_against_oracle(m, n, irow, icol, a, _vec(7, m), itnlim=8)
    assert s.info()["pat_wide"] == 601


def test_real32_wide_patterns():
    m, n, irow, icol, a, b = piecewise_mesh(240, 200, 8, 6)
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    s = lsqr_solver_ez().initialize(m, n, a32, irow, icol, itnlim=30, real32=True)
    assert s.info()["pat_wide"] > 256
    xp, yp = _vec(9, n).astype(np.float32), _vec(10, m).astype(np.float32)
    po = oracle.port()
    _, y_ref = po.aprod(1, m, n, irow, icol, a32.astype(np.float64), xp.astype(np.float64), yp.astype(np.float64))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, m, n, x, y)
    assert np.array_equal(y, y_ref.astype(np.float32))      # binary64 sums, rounded once
    os.environ["LSQRHIP_PAT"] = "0"
    s0 = lsqr_solver_ez().initialize(m, n, a32, irow, icol, itnlim=30, real32=True)
    assert s0.info()["sell"] != 3
    r, r0 = s.solve(b32, 0.0), s0.solve(b32, 0.0)
    assert np.array_equal(r.x, r0.x) and r.itn == r0.itn and r.anorm == r0.anorm


@pytest.mark.parametrize("name", sorted(CASES))
def test_the_golden_parity_cases_with_wide_patterns_forced(name):
    """... and with LSQRHIP_PAT2=1 beside LSQRHIP_PAT=1: systems of 257 ... 4096 rows, every row a pattern of its own, go
    through the wide table."""
    import test_gpu_parity as tp
    os.environ["LSQRHIP_PAT"] = "1"
    os.environ["LSQRHIP_PAT2"] = "1"
    tp.test_solve_parity_vs_reference_golden(name)


# ---------------------------------------------------------------------------------------------------------------------
# structure patterns (sell = 4): the column structure of the rows repeats, their values do not
# ---------------------------------------------------------------------------------------------------------------------
def variable(m, n, irow, icol, a, b, seed=5):
    """the same structure with arbitrary values"""
    return m, n, irow, icol, _vec(seed, a.size) * 2.0 - 1.0 + 0.25, b


def test_a_variable_coefficient_stencil_keeps_no_column_indices():
    p = P.poisson2d(300, 200)
    m, n, irow, icol, a, b = variable(p.m, p.n, p.irow, p.icol, p.a, p.b)
    s, r = check_against_oracle(m, n, irow, icol, a, b, want=4, want_t=4)
    info = s.info()
    nnzp = 5 * 64 * ((m + 63) // 64)
    assert info["csr_bytes"] <= 8 * nnzp + m + 4 * ((m + 63) // 64) + 8192      # values + a byte per row: no columns
    os.environ["LSQRHIP_SPAT"] = "0"
    s0 = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=25)
    assert s0.info()["sell"] == 1 and s0.info()["csr_bytes"] > info["csr_bytes"]
    r0 = s0.solve(b, 0.0)
    assert np.array_equal(r0.x, r.x) and r0.anorm == r.anorm and r0.rnorm == r.rnorm and r0.itn == r.itn


@pytest.mark.parametrize("case", ["3d7", "3d27", "holes", "conv31", "empty_rows", "unsym_dups"])
def test_structure_patterns_on_other_shapes(case):
    if case == "3d7":
        m, n, irow, icol, a, b, _ = grid3d(24, 19, 17, 1)
    elif case == "3d27":
        m, n, irow, icol, a, b, _ = grid3d(24, 19, 17, 3)
    elif case == "holes":
        drop = lambda r, k: ((k == 1) & (r % 7 == 0)) | ((k == 3) & (r % 11 == 0))   # noqa: E731
        m, n, irow, icol, a, b = stencil(9000, 9000, (-5, -1, 0, 1, 5), (1.0,) * 5, drop=drop)
    elif case == "conv31":
        m, n, irow, icol, a, b = stencil(30000, 30000, tuple(range(-15, 16)), (1.0,) * 31)
    elif case == "empty_rows":
        drop = lambda r, k: r % 5 == 2          # noqa: E731
        m, n, irow, icol, a, b = stencil(4000, 3990, (-1, 0, 2), (1.0,) * 3, drop=drop)
    else:
        m, n, irow, icol, a, b = stencil(5000, 5007, (3, -2, 0, 3, 1, -40), (1.0,) * 6, shuffle=True)
    m, n, irow, icol, a, b = variable(m, n, irow, icol, a, b)
    check_against_oracle(m, n, irow, icol, a, b, damp=1e-3, itnlim=10, want=4, want_t=4)


def test_structure_patterns_limits_and_real32():
    # random columns: no structure to speak of
    p = P.random_rows(30000, 20000, 8, seed=4)
    info = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol).info()
    assert info["sell"] not in (3, 4) and info["sell_t"] not in (3, 4)
    # a dictionary (three values) on 5 offsets: 3^5 = 243 interior rows and the boundary ones -- too many to number in a
    # byte, so no row patterns; and with a dictionary the packed records (3 bytes per nonzero) beat 8-byte values
    m, n, irow, icol, a, b = stencil(50000, 50000, (-3, -1, 0, 1, 3), (1.0,) * 5)
    a = np.choose((_vec(11, a.size) * 3).astype(int).clip(0, 2), [2.0, -1.0, 0.5])
    assert lsqr_solver_ez().initialize(m, n, a, irow, icol).info()["sell"] == 2
    # REAL32: float values and vectors, binary64 sums
    p = P.poisson2d(120, 90)
    m, n, irow, icol, a, b = variable(p.m, p.n, p.irow, p.icol, p.a, p.b)
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    s = lsqr_solver_ez().initialize(m, n, a32, irow, icol, itnlim=30, real32=True)
    assert s.info()["sell"] == 4
    xp, yp = _vec(9, n).astype(np.float32), _vec(10, m).astype(np.float32)
    _, y_ref = oracle.port().aprod(1, m, n, irow, icol, a32.astype(np.float64), xp.astype(np.float64), yp.astype(np.float64))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, m, n, x, y)
    assert np.array_equal(y, y_ref.astype(np.float32))
    os.environ["LSQRHIP_SPAT"] = "0"
    s0 = lsqr_solver_ez().initialize(m, n, a32, irow, icol, itnlim=30, real32=True)
    r, r0 = s.solve(b32, 0.0), s0.solve(b32, 0.0)
    assert np.array_equal(r.x, r0.x) and r.itn == r0.itn and r.anorm == r0.anorm


@pytest.mark.parametrize("name", sorted(CASES))
def test_the_golden_parity_cases_with_structure_patterns_forced(name):
    """... and once more with the value dictionary off and LSQRHIP_SPAT=1: whatever has <= 256 distinct column
    structures goes through k_spmv_spat."""
    import test_gpu_parity as tp
    os.environ.update(LSQRHIP_PAT="0", LSQRHIP_SPAT="1", LSQRHIP_VAL8="0")
    tp.test_solve_parity_vs_reference_golden(name)


@pytest.mark.parametrize("layout", [{}, {"LSQRHIP_PAT": "0"}, {"LSQRHIP_PAT": "0", "LSQRHIP_SELLP": "0"},
                                    {"LSQRHIP_PAT": "0", "LSQRHIP_VAL8": "0"},
                                    {"LSQRHIP_PAT": "0", "LSQRHIP_VAL8": "0", "LSQRHIP_SPAT": "0"}])
def test_non_temporal_streams_change_no_bit(layout):
    """The stream policy (common.h ld_stream: plain loads while an iteration's working set fits the Infinity Cache,
    non-temporal beyond) selects another instantiation of the same kernel: every short-row layout must give the same
    bits either way."""
    p = P.poisson2d(200, 150)
    os.environ.update(layout)
    outs = []
    for nt in ("0", "1"):
        os.environ["LSQRHIP_STREAM_NT"] = nt
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=30)
        x, y = _vec(9, p.n), _vec(10, p.m)
        s.aprod(1, p.m, p.n, x, y)
        outs.append((s.info()["sell"], y, s.solve(p.b, 1e-3)))
    (l0, y0, r0), (l1, y1, r1) = outs
    assert l0 == l1 != 0
    assert np.array_equal(y0, y1) and np.array_equal(r0.x, r1.x) and r0.anorm == r1.anorm and r0.itn == r1.itn
