"""Build a solver whose matrix is generated directly in HBM (csrc/gen_api.h).

The generated triplets are bit-identical to lsqr_amd.problems (same counter-based hash), so
small instances can be cross-checked on the host while the 10^8..10^9-nonzero benchmark
configurations never touch PCIe.  `row0 / nrows` select a row block of the global system
(the shard of one rank).
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass

import numpy as np

from . import capi, problems as P
from .capi import DeviceBuffer, check, lib
from .solver import lsqr_solver_ez

KIND_RANDOM, KIND_POISSON, KIND_BY_ROWPTR, KIND_MESH = 0, 1, 2, 3


@dataclass
class DeviceProblem:
    name: str
    m: int            # global rows
    n: int            # columns
    row0: int         # first global row of this block
    nrows: int        # rows in this block
    nnz: int          # local nonzeros
    solver: lsqr_solver_ez
    d_b: DeviceBuffer  # b for the local rows, in HBM
    damp: float = 0.0


def parse_spec(spec: str):
    """'poisson2d:NX:NY' | 'mesh2d:NX:NY:BX:BY' | 'random:M:N:PER_ROW' | 'powerlaw:M:N:DMAX[:DMIN]' -> dict"""
    kind, *a = spec.split(":")
    a = [int(t) for t in a]
    if kind == "poisson2d":
        return dict(kind="poisson2d", m=a[0] * a[1], n=a[0] * a[1], nx=a[0], ny=a[1], damp=0.0)
    if kind == "mesh2d":
        return dict(kind="mesh2d", m=a[0] * a[1], n=a[0] * a[1], nx=a[0], ny=a[1], bx=a[2], by=a[3], damp=0.0)
    if kind == "random":
        return dict(kind="random", m=a[0], n=a[1], per_row=a[2], damp=1e-3)
    if kind == "powerlaw":
        return dict(kind="powerlaw", m=a[0], n=a[1], dmax=a[2], dmin=a[3] if len(a) > 3 else 4, damp=0.0)
    raise ValueError(f"unknown workload {spec}")


def row_weights(cfg: dict, seed: int = 12345) -> np.ndarray | None:
    """Per-row nonzero counts when they are not uniform (used by the nnz-balanced partitioner)."""
    if cfg["kind"] == "powerlaw":
        return P.powerlaw_degrees(cfg["m"], seed, cfg["dmin"], cfg["dmax"])
    return None


def generate(spec: str, row0: int = 0, nrows: int | None = None, seed: int = 12345, **solver_kw) -> DeviceProblem:
    cfg = parse_spec(spec)
    m, n = cfg["m"], cfg["n"]
    nrows = m - row0 if nrows is None else nrows
    L = lib()
    d_rowptr = None
    if cfg["kind"] == "random":
        kind, p0, p1 = KIND_RANDOM, cfg["per_row"], 0
        nnz = nrows * cfg["per_row"]
    elif cfg["kind"] == "poisson2d":
        kind, p0, p1 = KIND_POISSON, cfg["nx"], cfg["ny"]
        nnz = int(L.lsqrhip_gen_count(kind, m, n, p0, p1, row0, nrows))
    elif cfg["kind"] == "mesh2d":
        kind, p0, p1 = KIND_MESH, cfg["nx"], (cfg["bx"] << 16) | cfg["by"]
        nnz = int(L.lsqrhip_gen_count(kind, m, n, p0, p1, row0, nrows))
    else:
        kind, p0, p1 = KIND_BY_ROWPTR, cfg["dmin"], cfg["dmax"]
        deg = P.powerlaw_degrees(m, seed, cfg["dmin"], cfg["dmax"], row0=row0, nrows=nrows)
        ptr = np.zeros(nrows + 1, dtype=np.int64)
        np.cumsum(deg, out=ptr[1:])
        nnz = int(ptr[-1])
        d_rowptr = DeviceBuffer.from_array(ptr)
    d_irow = DeviceBuffer(4 * max(nnz, 1))
    d_icol = DeviceBuffer(4 * max(nnz, 1))
    d_a = DeviceBuffer(8 * max(nnz, 1))
    if cfg["kind"] == "poisson2d":
        k = np.arange(row0, row0 + nrows, dtype=np.float64)
        d_b = DeviceBuffer.from_array(np.sin(0.001 * (k + 1.0)))     # host libm, like problems.poisson2d
        pb = None
    else:
        d_b = DeviceBuffer(8 * max(nrows, 1))
        pb = d_b.ptr
    out = C.c_int64()
    check(L.lsqrhip_gen_coo(kind, seed, m, n, p0, p1, row0, nrows, d_rowptr.ptr if d_rowptr else None,
                            d_irow.ptr, d_icol.ptr, d_a.ptr, pb, C.addressof(out)))
    assert out.value == nnz, (out.value, nnz)
    check(L.lsqrhip_dev_sync())
    t_build = time.perf_counter()
    s = lsqr_solver_ez().initialize_from_device_coo(nrows, n, nnz, d_irow.ptr, d_icol.ptr, d_a.ptr, **solver_kw)
    s.build_seconds = time.perf_counter() - t_build      # initialize: device COO -> ready-to-solve layouts
    for buf in (d_irow, d_icol, d_a, d_rowptr):
        if buf is not None:
            buf.free()
    return DeviceProblem(spec, m, n, row0, nrows, nnz, s, d_b, cfg["damp"])


def download_coo(spec: str, row0: int = 0, nrows: int | None = None, seed: int = 12345):
    """(irow, icol, a, b) of a device-generated block, copied to the host (tests only use this)."""
    cfg = parse_spec(spec)
    m, n = cfg["m"], cfg["n"]
    nrows = m - row0 if nrows is None else nrows
    L = lib()
    d_rowptr = None
    if cfg["kind"] == "random":
        kind, p0, p1, nnz = KIND_RANDOM, cfg["per_row"], 0, nrows * cfg["per_row"]
    elif cfg["kind"] == "poisson2d":
        kind, p0, p1 = KIND_POISSON, cfg["nx"], cfg["ny"]
        nnz = int(L.lsqrhip_gen_count(kind, m, n, p0, p1, row0, nrows))
    elif cfg["kind"] == "mesh2d":
        kind, p0, p1 = KIND_MESH, cfg["nx"], (cfg["bx"] << 16) | cfg["by"]
        nnz = int(L.lsqrhip_gen_count(kind, m, n, p0, p1, row0, nrows))
    else:
        kind, p0, p1 = KIND_BY_ROWPTR, cfg["dmin"], cfg["dmax"]
        deg = P.powerlaw_degrees(m, seed, cfg["dmin"], cfg["dmax"], row0=row0, nrows=nrows)
        ptr = np.zeros(nrows + 1, dtype=np.int64)
        np.cumsum(deg, out=ptr[1:])
        nnz = int(ptr[-1])
        d_rowptr = DeviceBuffer.from_array(ptr)
    d_irow, d_icol, d_a = DeviceBuffer(4 * max(nnz, 1)), DeviceBuffer(4 * max(nnz, 1)), DeviceBuffer(8 * max(nnz, 1))
    d_b = DeviceBuffer(8 * max(nrows, 1))
    out = C.c_int64()
    check(L.lsqrhip_gen_coo(kind, seed, m, n, p0, p1, row0, nrows, d_rowptr.ptr if d_rowptr else None,
                            d_irow.ptr, d_icol.ptr, d_a.ptr, None if kind == KIND_POISSON else d_b.ptr,
                            C.addressof(out)))
    return (d_irow.to_array(np.int32, nnz), d_icol.to_array(np.int32, nnz), d_a.to_array(np.float64, nnz),
            None if kind == KIND_POISSON else d_b.to_array(np.float64, nrows))
