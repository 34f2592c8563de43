"""lsqr_amd/problems.py mesh2d -- the five-point operator whose coefficient is constant on each of bx x by regions (the
workload of the wide row patterns, csrc/pat.h; its device twin is csrc/gen_api.h k_gen_mesh, held to it bit for bit by
tests/test_gpu_devgen.py): the matrix is what its docstring says, and the oracle solves it as scipy does."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

import oracle
from lsqr_amd import problems as P


def test_the_mesh_is_a_symmetric_m_matrix_with_one_coefficient_per_region():
    nx, ny, bx, by = 61, 47, 5, 3
    p = P.mesh2d(nx, ny, bx, by)
    A = sp.coo_matrix((p.a, (p.irow - 1, p.icol - 1)), shape=(p.m, p.n)).tocsr()
    assert p.m == p.n == nx * ny and p.nnz == 5 * nx * ny - 2 * nx - 2 * ny
    assert abs(A - A.T).max() == 0.0                                     # harmonic means: the same from both sides
    assert (A.diagonal() > 0).all() and (A - sp.diags(A.diagonal())).max() <= 0.0
    assert (A.sum(axis=1) >= -1e-12).all()                               # weakly diagonally dominant (Dirichlet faces)
    # the order inside a row is by ascending column, rows in order (what the device generator emits)
    assert (np.diff(p.irow) >= 0).all()
    same = np.diff(p.irow) == 0
    assert (np.diff(p.icol)[same] > 0).all()
    # a cell strictly inside a region sees its own k four times: diagonal 4k, off-diagonals -k, k in (1, 3)
    c = (ny // (2 * by)) * nx + nx // (2 * bx)
    row = A.getrow(c)
    k = row.data.max() / 4.0
    assert 1.0 < k < 3.0 and sorted(row.data)[:4] == [-k] * 4
    # one coefficient per region: the cells strictly inside any region have bx * by distinct diagonals
    i, j = np.arange(p.m) % nx, np.arange(p.m) // nx
    region = (i * bx) // nx + bx * ((j * by) // ny)
    inner = np.ones(p.m, bool)
    for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        ii, jj = i + di, j + dj
        ok = (ii >= 0) & (ii < nx) & (jj >= 0) & (jj < ny)
        inner &= ok & (region[np.where(ok, ii + nx * jj, 0)] == region)
    assert len(np.unique(A.diagonal()[inner])) == bx * by


def test_distinct_rows_of_the_bench_shape_scaled_down():
    """12 x 10 regions: interiors, interfaces, corners and the boundary of the grid make ~10^3 distinct rows -- more
    than the one-byte pattern table holds, fewer than the wide one's 4096."""
    p = P.mesh2d(500, 400, 12, 10, seed=11)
    key = np.zeros(p.m, dtype=np.uint64)
    h = (p.icol.astype(np.int64) - p.irow.astype(np.int64)).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) ^ p.a.view(np.uint64)
    with np.errstate(over="ignore"):
        np.add.at(key, p.irow - 1, h * (h >> np.uint64(7) | np.uint64(1)))
    assert 256 < len(np.unique(key)) <= 4096


def test_the_oracle_solves_the_mesh_as_scipy_does():
    p = P.mesh2d(40, 30, 4, 3)
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=0.0, atol=1e-12, btol=1e-12, itnlim=2000)
    A = sp.coo_matrix((p.a, (p.irow - 1, p.icol - 1)), shape=(p.m, p.n)).tocsr()
    x = sla.spsolve(A.tocsc(), p.b)
    assert o.istop in (1, 2) and np.linalg.norm(o.x - x) <= 1e-8 * np.linalg.norm(x)
