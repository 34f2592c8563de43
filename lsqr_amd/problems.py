"""Synthetic sparse systems for the configurations of BASELINE.json.

Everything is produced from a counter-based hash RNG (splitmix64 of
``(seed, stream, i, t)``) in pure integer arithmetic, so the host generators
here and the on-device generators in ``csrc/gen.hip`` emit bit-identical
``(irow, icol, a, b)``.  All COO output is 1-based, like the reference's
``lsqr_solver_ez%initialize`` expects (reference src/lsqr.f90:91-118).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_U = np.uint64
_GOLD = _U(0x9E3779B97F4A7C15)
_M1 = _U(0xBF58476D1CE4E5B9)
_M2 = _U(0x94D049BB133111EB)
_STREAM = _U(0x632BE59BD9B4E019)

# stream ids (shared with csrc/gen.hip)
S_COL, S_VAL, S_B, S_DEG, S_PERM = 1, 2, 3, 4, 5


def _sm64(z):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    z = (z + _GOLD).astype(np.uint64)
    z = ((z ^ (z >> _U(30))) * _M1).astype(np.uint64)
    z = ((z ^ (z >> _U(27))) * _M2).astype(np.uint64)
    return z ^ (z >> _U(31))


def rng_u64(seed: int, stream: int, i, t=0):
    """h(seed, stream, i, t): the one hash both host and device use."""
    with np.errstate(over="ignore"):
        i = np.asarray(i, dtype=np.uint64)
        t = np.asarray(t, dtype=np.uint64)
        h = _sm64(np.asarray(_U(seed) ^ (_U(stream) * _STREAM), dtype=np.uint64))
        h = _sm64(h ^ i)
        h = _sm64(h ^ t)
    return h


def u64_to_index(h, n: int):
    """Uniform integer in [0, n) from the top 32 bits (n < 2**31)."""
    return ((h >> _U(32)) * _U(n)) >> _U(32)


def u64_to_unit(h):
    """Uniform double in (-1, 1): exact in binary64 on host and device."""
    return (h >> _U(11)).astype(np.float64) * (2.0 ** -52) - 1.0


@dataclass
class Problem:
    name: str
    m: int
    n: int
    irow: np.ndarray  # int32, 1-based
    icol: np.ndarray  # int32, 1-based
    a: np.ndarray     # float64
    b: np.ndarray     # float64 [m]
    damp: float = 0.0

    @property
    def nnz(self) -> int:
        return int(self.a.size)

    def dense(self) -> np.ndarray:
        A = np.zeros((self.m, self.n))
        np.add.at(A, (self.irow - 1, self.icol - 1), self.a)
        return A


# ---------------------------------------------------------------------------
# config 1: the reference's own toy systems
# ---------------------------------------------------------------------------

def readme_3x3() -> Problem:
    """README.md:33-38 / test/lsqrtest_ez.f90:20-27 (test_1)."""
    icol = np.array([1, 1, 1, 2, 2, 2, 3, 3, 3], dtype=np.int32)
    irow = np.array([1, 2, 3, 1, 2, 3, 1, 2, 3], dtype=np.int32)
    a = np.array([1, 4, 7, 2, 5, 88, 3, 66, 9], dtype=np.float64)
    return Problem("readme_3x3", 3, 3, irow, icol, a, np.array([1.0, 2.0, 3.0]))


def ez_3x4() -> Problem:
    """test/lsqrtest_ez.f90:70-78 (test_2, under-determined)."""
    icol = np.array([1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4], dtype=np.int32)
    irow = np.array([1, 2, 3] * 4, dtype=np.int32)
    a = np.array([4.1, 1.1, 11.1, 5.1, -3.1, 3.1, 66.1, 8.1, -87.1, 0.1, -9.1, 2.1])
    return Problem("ez_3x4", 3, 4, irow, icol, a, np.array([1.0, 2.0, 3.0]))


# ---------------------------------------------------------------------------
# config 2: 5-point Poisson on an nx-by-ny grid (square, damp = 0)
# ---------------------------------------------------------------------------

def poisson2d(nx: int, ny: int) -> Problem:
    """Row k = j*nx + i (0-based): 4 on the diagonal, -1 at (i-1,j), (i+1,j),
    (i,j-1), (i,j+1) where inside the grid.  Entries are emitted row by row in
    ascending column order; b(k) = sin(0.001*(k+1)).  nx=ny=1000 is config 2
    (nnz = 4 996 000)."""
    N = nx * ny
    k = np.arange(N, dtype=np.int64)
    i = k % nx
    j = k // nx
    cols = np.stack([k - nx, k - 1, k, k + 1, k + nx], axis=1)
    vals = np.tile(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), (N, 1))
    keep = np.stack([j > 0, i > 0, np.ones(N, bool), i < nx - 1, j < ny - 1], axis=1)
    rows = np.repeat(k[:, None], 5, axis=1)
    irow = (rows[keep] + 1).astype(np.int32)
    icol = (cols[keep] + 1).astype(np.int32)
    a = vals[keep]
    b = np.sin(0.001 * (k + 1).astype(np.float64))
    return Problem(f"poisson2d_{nx}x{ny}", N, N, irow, icol, a, b)


# ---------------------------------------------------------------------------
# a five-point mesh whose coefficient is constant on each of bx x by regions (no BASELINE configuration: the matrix
# of a few thousand distinct rows that the wide row patterns of csrc/pat.h are for)
# ---------------------------------------------------------------------------

def mesh2d(nx: int, ny: int, bx: int, by: int, seed: int = 12345) -> Problem:
    """-div(k grad u) on an nx x ny grid: row c = j*nx + i holds -kS, -kW, kS + kW + kE + kN, -kE, -kN (ascending
    columns, a face outside the grid left out of the row but kept in the diagonal with the cell's own k: Dirichlet);
    k of a cell = 2 + u(-1, 1) of its region ((i*bx)//nx, (j*by)//ny), k of a face = the harmonic mean 2 k1 k2 / (k1 + k2)
    of its two cells; b = u(-1, 1).  Same hash and the same operation order as the device generator (csrc/gen_api.h)."""
    N = nx * ny
    c = np.arange(N, dtype=np.int64)
    i, j = c % nx, c // nx
    region = (i * bx) // nx + bx * ((j * by) // ny)
    kreg = 2.0 + u64_to_unit(rng_u64(seed, S_VAL, np.arange(bx * by, dtype=np.uint64)))
    k = kreg[region]

    def face(ok, nb):
        kn = k[np.where(ok, nb, 0)]
        return np.where(ok, 2.0 * k * kn / (k + kn), k)

    inside = [j > 0, i > 0, i < nx - 1, j < ny - 1]
    nbr = [c - nx, c - 1, c + 1, c + nx]
    f = [face(ok, nb) for ok, nb in zip(inside, nbr)]
    diag = ((f[0] + f[1]) + f[2]) + f[3]
    cols = np.stack([nbr[0], nbr[1], c, nbr[2], nbr[3]], axis=1)
    vals = np.stack([-f[0], -f[1], diag, -f[2], -f[3]], axis=1)
    keep = np.stack([inside[0], inside[1], np.ones(N, bool), inside[2], inside[3]], axis=1)
    rows = np.repeat(c[:, None], 5, axis=1)
    b = u64_to_unit(rng_u64(seed, S_B, c.astype(np.uint64)))
    return Problem(f"mesh2d_{nx}x{ny}_{bx}x{by}", N, N, (rows[keep] + 1).astype(np.int32),
                   (cols[keep] + 1).astype(np.int32), vals[keep], b)


# ---------------------------------------------------------------------------
# configs 3/4: random rectangular, fixed nnz per row (duplicates allowed)
# ---------------------------------------------------------------------------

def random_rows(m: int, n: int, per_row: int, seed: int = 12345, damp: float = 0.0,
                row0: int = 0, nrows: int | None = None) -> Problem:
    """Row k holds `per_row` draws: column uniform in [0,n) (duplicates kept, the
    reference sums them, src/lsqr.f90:168-172), value uniform(-1,1), b uniform(-1,1).
    `row0/nrows` generate a row block of the same global matrix (multi-GPU shards)."""
    nrows = m - row0 if nrows is None else nrows
    k = np.arange(row0, row0 + nrows, dtype=np.uint64)
    t = np.arange(per_row, dtype=np.uint64)
    kk = np.repeat(k, per_row)
    tt = np.tile(t, nrows)
    icol = (u64_to_index(rng_u64(seed, S_COL, kk, tt), n) + _U(1)).astype(np.int32)
    a = u64_to_unit(rng_u64(seed, S_VAL, kk, tt))
    irow = (kk + _U(1)).astype(np.int32)
    b = u64_to_unit(rng_u64(seed, S_B, k))
    return Problem(f"random_{m}x{n}_r{per_row}", m, n, irow, icol, a, b, damp)


# ---------------------------------------------------------------------------
# config 5: power-law row degrees
# ---------------------------------------------------------------------------

def powerlaw_cdf(dmin: int, dmax: int, gamma: float = 2.1) -> np.ndarray:
    """Integer CDF table T[d-dmin] = floor(2**64 * P(D <= d)) clipped to uint64,
    with P(d) ~ d**-gamma on [dmin, dmax].  Host builds it once; host and device
    then draw degrees by the same integer binary search."""
    d = np.arange(dmin, dmax + 1, dtype=np.float64)
    p = d ** (-gamma)
    c = np.cumsum(p) / np.sum(p)
    tbl = np.minimum(np.floor(c * 2.0 ** 64), 2.0 ** 64 - 2048.0).astype(np.uint64)
    tbl[-1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return tbl


def powerlaw_degrees(m: int, seed: int, dmin: int, dmax: int, gamma: float = 2.1,
                     row0: int = 0, nrows: int | None = None) -> np.ndarray:
    nrows = m - row0 if nrows is None else nrows
    tbl = powerlaw_cdf(dmin, dmax, gamma)
    k = np.arange(row0, row0 + nrows, dtype=np.uint64)
    h = rng_u64(seed, S_DEG, k)
    deg = np.searchsorted(tbl, h, side="left").astype(np.int64) + dmin
    deg = np.minimum(deg, dmax)
    # force at least one row at the maximum degree
    forced = int(u64_to_index(rng_u64(seed, S_PERM, np.uint64(0)), m))
    if row0 <= forced < row0 + nrows:
        deg[forced - row0] = dmax
    return deg


def powerlaw_rows(m: int, n: int, seed: int = 12345, dmin: int = 4, dmax: int = 10000,
                  gamma: float = 2.1, damp: float = 0.0) -> Problem:
    """Truncated discrete power-law row degrees, uniform columns, uniform values.
    Degrees are an i.i.d. hash of the row index, i.e. already 'randomly permuted'."""
    deg = powerlaw_degrees(m, seed, dmin, dmax, gamma)
    ptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(deg, out=ptr[1:])
    nnz = int(ptr[-1])
    kk = np.repeat(np.arange(m, dtype=np.uint64), deg)
    tt = (np.arange(nnz, dtype=np.int64) - np.repeat(ptr[:-1], deg)).astype(np.uint64)
    icol = (u64_to_index(rng_u64(seed, S_COL, kk, tt), n) + _U(1)).astype(np.int32)
    a = u64_to_unit(rng_u64(seed, S_VAL, kk, tt))
    irow = (kk + _U(1)).astype(np.int32)
    b = u64_to_unit(rng_u64(seed, S_B, np.arange(m, dtype=np.uint64)))
    return Problem(f"powerlaw_{m}x{n}_d{dmin}-{dmax}", m, n, irow, icol, a, b, damp)


# ---------------------------------------------------------------------------
# helpers for edge-case tests
# ---------------------------------------------------------------------------

def shuffled(p: Problem, seed: int = 7) -> Problem:
    """Same matrix, COO triplets in a scrambled order (the reference accepts any
    order; it only changes the summation order, src/lsqr.f90:168-172)."""
    key = rng_u64(seed, S_PERM, np.arange(p.nnz, dtype=np.uint64), 1)
    perm = np.argsort(key, kind="stable")
    return Problem(p.name + "_shuffled", p.m, p.n, p.irow[perm].copy(), p.icol[perm].copy(),
                   p.a[perm].copy(), p.b.copy(), p.damp)


def checksum(p: Problem) -> str:
    import hashlib
    h = hashlib.sha256()
    for arr in (p.irow, p.icol, p.a, p.b):
        h.update(np.ascontiguousarray(arr).tobytes())
    return h.hexdigest()[:16]
