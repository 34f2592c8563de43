"""Host-side mirror of the reference's `lsqr_solver_ez` type over the C-ABI.

Same names, argument meaning and error behaviour as reference src/lsqr.f90:32-65:

    solver = lsqr_solver_ez()
    solver.initialize(m, n, a, irow, icol, atol=..., btol=..., conlim=..., itnlim=..., nout=...)
    r = solver.solve(b, damp)            # r.x, r.istop, r.itn, r.anorm, ...
    solver.aprod(mode, m, n, x, y)       # in place, like the reference

Where the reference `error stop`s with a fixed string, this raises
`LsqrHipError` carrying that exact string (`.message`).  All arithmetic happens
in liblsqrhip.so on the GPU; this file only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import io
import sys
from dataclasses import dataclass

import numpy as np

from . import capi
from .capi import LsqrHipError, Timing, check, lib
from .logfmt import format_log

_EPS = float(np.finfo(np.float64).eps)


@dataclass
class SolveResult:
    """Outputs of `solve` (reference src/lsqr.f90:215-223)."""
    x: np.ndarray
    istop: int
    itn: int
    anorm: float
    acond: float
    rnorm: float
    arnorm: float
    xnorm: float
    se: np.ndarray | None = None


class lsqr_solver_ez:  # noqa: N801  (name kept from the reference)
    """GPU-resident twin of the reference's `lsqr_solver_ez` (src/lsqr.f90:32-65)."""

    def __init__(self):
        self._h = C.c_void_p()
        self._reset()

    # -- reference component defaults, src/lsqr.f90:39-51 ---------------------
    def _reset(self):
        self.m = 0
        self.n = 0
        self.num_nonzero_elements = 0
        self.atol = 0.0
        self.btol = 0.0
        self.conlim = 0.0
        self.itnlim = 100
        self.nout = 0
        self.real32 = False

    def _free(self):
        if getattr(self, "_h", None):
            lib().lsqrhip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    # -- initialize_ez, src/lsqr.f90:91-127 ------------------------------------
    def initialize(self, m, n, a, irow, icol, atol=None, btol=None, conlim=None, itnlim=None, nout=None,
                   real32=False):
        """`me` is intent(out) in the reference (:95): every call starts from the defaults.

        real32=True is the reference's REAL32 build (src/lsqr_kinds.F90:16-17: wp = real32): a, b, x, se
        are float32 here and on the device (binary64 in registers only)."""
        self._free()
        self._reset()
        a = np.ascontiguousarray(a, dtype=np.float32 if real32 else np.float64)
        irow = np.ascontiguousarray(irow, dtype=np.int32)
        icol = np.ascontiguousarray(icol, dtype=np.int32)
        if not (a.size == irow.size == icol.size):                      # :109
            raise LsqrHipError(capi.ERR_SIZES, lib().lsqrhip_error_string(capi.ERR_SIZES).decode())
        h = C.c_void_p()
        create = lib().lsqrhip_create_f32 if real32 else lib().lsqrhip_create
        check(create(int(m), int(n), a.size, irow.ctypes.data, icol.ctypes.data, a.ctypes.data, C.byref(h)))  # :110-118
        self._h = h
        self.real32 = bool(real32)
        self.m, self.n, self.num_nonzero_elements = int(m), int(n), int(a.size)
        if atol is not None:
            self.atol = float(atol)                                     # :121-125
        if btol is not None:
            self.btol = float(btol)
        if conlim is not None:
            self.conlim = float(conlim)
        if itnlim is not None:
            self.itnlim = int(itnlim)
        if nout is not None:
            self.nout = nout
        return self

    def initialize_from_device_coo(self, m, n, nnz, d_irow, d_icol, d_a, atol=None, btol=None, conlim=None,
                                   itnlim=None, nout=None):
        """`initialize` for triplets that already live in HBM (1-based; raw device addresses)."""
        self._free()
        self._reset()
        h = C.c_void_p()
        check(lib().lsqrhip_create_from_device_coo(int(m), int(n), int(nnz), d_irow, d_icol, d_a, C.byref(h)))
        self._h = h
        self.m, self.n, self.num_nonzero_elements = int(m), int(n), int(nnz)
        for k, v in (("atol", atol), ("btol", btol), ("conlim", conlim)):
            if v is not None:
                setattr(self, k, float(v))
        if itnlim is not None:
            self.itnlim = int(itnlim)
        if nout is not None:
            self.nout = nout
        return self

    def _need(self):
        if not self._h:
            raise LsqrHipError(capi.ERR_NOT_INIT, lib().lsqrhip_error_string(capi.ERR_NOT_INIT).decode())

    # -- solve_ez, src/lsqr.f90:207-259 ----------------------------------------
    def solve(self, b, damp=0.0, wantse=False) -> SolveResult:
        self._need()
        wp = np.float32 if self.real32 else np.float64
        b = np.ascontiguousarray(b, dtype=wp)
        if b.shape != (self.m,):
            raise ValueError(f"b must have shape ({self.m},)")          # explicit-shape dummy b(me%m), :213
        x = np.zeros(max(self.n, 1), dtype=wp)
        se = np.zeros(max(self.n, 1), dtype=wp) if wantse else None
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        want_log = 1 if self.nout else 0
        solve = lib().lsqrhip_solve_f32 if self.real32 else lib().lsqrhip_solve
        check(solve(self._h, b.ctypes.data, float(damp), self.atol, self.btol, self.conlim,
                    self.itnlim, int(bool(wantse)), want_log, x.ctypes.data,
                    se.ctypes.data if wantse else None, C.addressof(istop), C.addressof(itn),
                    *[C.addressof(s) for s in sc]))
        r = SolveResult(x[:self.n], istop.value, itn.value, *[s.value for s in sc],
                        se=se[:self.n] if wantse else None)
        if self.nout:
            self._write_log(r, float(damp), bool(wantse))
        return r

    def solve_device(self, d_b: int, d_x: int, damp=0.0, d_se: int | None = None) -> SolveResult:
        """b, x (and se) already resident in HBM: raw device addresses (no PCIe in the call)."""
        self._need()
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        check(lib().lsqrhip_solve_device(self._h, d_b, float(damp), self.atol, self.btol, self.conlim,
                                         self.itnlim, 1 if d_se else 0, 0, d_x, d_se,
                                         C.addressof(istop), C.addressof(itn), *[C.addressof(s) for s in sc]))
        return SolveResult(np.zeros(0), istop.value, itn.value, *[s.value for s in sc])

    # -- aprod_ez, src/lsqr.f90:134-200 ----------------------------------------
    def aprod(self, mode, m, n, x, y):
        """mode 1: y += A x ; mode 2: x += A' y.  x, y are float64 (real32 build: float32) numpy arrays
        updated in place."""
        self._need()
        if m != self.m or n != self.n:                                  # :152
            raise LsqrHipError(capi.ERR_NOT_INIT, lib().lsqrhip_error_string(capi.ERR_NOT_INIT).decode())
        wp = np.float32 if self.real32 else np.float64
        for v, k in ((x, self.n), (y, self.m)):
            if not (isinstance(v, np.ndarray) and v.dtype == wp and v.flags.c_contiguous and v.size == k):
                raise ValueError(f"x, y must be contiguous {np.dtype(wp).name} arrays of length n, m")
        aprod = lib().lsqrhip_aprod_f32 if self.real32 else lib().lsqrhip_aprod
        check(aprod(self._h, int(mode), x.ctypes.data, y.ctypes.data))  # :197 -> ERR_MODE

    # -- acheck / xcheck, src/lsqr.f90:908-994, 1015-1154 ------------------------
    def acheck(self, eps=None):
        """eps defaults to the machine precision of the build's working precision, as the callers of the reference
        pass it (test/lsqrtest_module.f90:184)."""
        self._need()
        if eps is None:
            eps = float(np.finfo(np.float32).eps) if self.real32 else _EPS
        inform, err = C.c_int(), C.c_double()
        fn = lib().lsqrhip_acheck_f32 if self.real32 else lib().lsqrhip_acheck
        check(fn(self._h, float(eps), C.addressof(inform), C.addressof(err)))
        return inform.value, err.value

    def xcheck(self, anorm, damp, b, x, eps=None):
        self._need()
        wp = np.float32 if self.real32 else np.float64
        if eps is None:
            eps = float(np.finfo(wp).eps)
        b = np.ascontiguousarray(b, dtype=wp)
        x = np.ascontiguousarray(x, dtype=wp)
        u, v, w = np.zeros(max(self.m, 1), wp), np.zeros(max(self.n, 1), wp), np.zeros(max(self.n, 1), wp)
        tests = np.zeros(3)
        inform = C.c_int()
        fn = lib().lsqrhip_xcheck_f32 if self.real32 else lib().lsqrhip_xcheck
        check(fn(self._h, float(anorm), float(damp), float(eps), b.ctypes.data, x.ctypes.data,
                                   u.ctypes.data, v.ctypes.data, w.ctypes.data, C.addressof(inform),
                                   tests.ctypes.data))
        return inform.value, tests, u[:self.m], v[:self.n], w[:self.n]

    # -- measurement / options --------------------------------------------------
    def set_option(self, name: str, value: int):
        self._need()
        check(lib().lsqrhip_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        self._need()
        v = C.c_int64()
        check(lib().lsqrhip_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def bench_kernel(self, which: int, reps: int) -> float:
        """Average ms of `reps` back-to-back launches of hot kernel 1 (mode-1 SpMV), 2 (mode-2 SpMV)
        or 3 (x/w update), one HIP event pair around the whole run."""
        self._need()
        ms = C.c_double()
        check(lib().lsqrhip_bench_kernel(self._h, int(which), int(reps), C.addressof(ms)))
        return ms.value

    def last_timing(self) -> Timing:
        self._need()
        t = Timing()
        check(lib().lsqrhip_last_timing(self._h, C.byref(t)))
        return t

    def info(self) -> dict:
        self._need()
        d = (C.c_int64 * 16)()
        check(lib().lsqrhip_info(self._h, d))
        return dict(m=d[0], n=d[1], nnz=d[2], csr_bytes=d[3], csrt_bytes=d[4], rowptr_bytes=d[5],
                    dict_entries=d[6], value_bytes=d[7], col_bytes=d[8], colt_bytes=d[9],
                    panels=d[10], panels_t=d[11], sell=d[12], sell_t=d[13], xlds=d[14], xlds_t=d[15],
                    # patterns of the wide row-pattern table (sell = 3 with two-byte pattern numbers; 0: not in use)
                    pat_wide=self.get_option("pat_wide_mode1"), pat_wide_t=self.get_option("pat_wide_mode2"),
                    # row patterns in the paired-rows form (lane L owns rows 2L, 2L + 1: csrc/pat.h)
                    pat_pair=self.get_option("pat_pair_mode1"), pat_pair_t=self.get_option("pat_pair_mode2"))

    def log_records(self) -> np.ndarray:
        self._need()
        k = lib().lsqrhip_log_count(self._h)
        rec = np.zeros((k, capi.LOG_STRIDE))
        if k:
            check(lib().lsqrhip_log_fetch(self._h, 0, k, rec.ctypes.data))
        return rec

    def _write_log(self, r: SolveResult, damp: float, wantse: bool):
        ex = np.zeros(6)
        check(lib().lsqrhip_log_extras(self._h, ex.ctypes.data))
        text = format_log(self.m, self.n, damp, wantse, self.atol, self.btol, self.conlim, self.itnlim,
                          self.log_records(), r, bnorm=ex[0], dxmax=ex[1], maxdx=int(ex[2]), test2_0=ex[5],
                          beta0=ex[4])
        out = self.nout
        if isinstance(out, (io.IOBase,)) or hasattr(out, "write"):
            out.write(text)
        elif isinstance(out, str):
            with open(out, "w") as f:
                f.write(text)
        else:  # any non-zero unit number: standard output, like output_unit
            sys.stdout.write(text)
