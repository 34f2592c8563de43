"""LSQR on a user-supplied DEVICE operator, and the reference's own test-problem class.

`lsqr_solver_device` mirrors the reference's abstract class `lsqr_solver`
(src/lsqr.f90:16-30: deferred `aprod`, public `lsqr`, `acheck`, `xcheck`) for operators that
live on the GPU: subclass it and override `aprod_device`, exactly as
test/lsqrtest_module.f90:35-44 subclasses `lsqr_solver` and overrides `aprod` -- except that
x and y arrive as device addresses plus a stream, and the override only enqueues work.

`saunders_problem` is that reference subclass itself: the operator A = HY*D*HZ with the problem
generator `lstp` (test/lsqrtest_module.f90:283-505), built into liblsqrhip.so as a device
operator.  `run_suite` is `lsqr_test` (test/lsqrtest_module.f90:55-94): the 18 problems, each
through acheck -> LSQR -> xcheck -> error against xtrue, logged in the layout of LSQR.LIS.

All arithmetic happens in liblsqrhip.so on the GPU; this file marshals arguments and prints.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import check, lib
from .logfmt import fE, fI
from .solver import SolveResult, lsqr_solver_ez

_EPS = float(np.finfo(np.float64).eps)


class lsqr_solver_device(lsqr_solver_ez):  # noqa: N801  (named after the reference's lsqr_solver)
    """User operator on the device.  Override `aprod_device`, or pass `aprod=` a callable with
    the same signature.  `solve`, `aprod`, `acheck`, `xcheck`, `nout` logging work as on
    `lsqr_solver_ez`; `lsqr(...)` takes the reference's argument list (src/lsqr.f90:432-435)."""

    def initialize(self, m, n, aprod=None, atol=None, btol=None, conlim=None, itnlim=None, nout=None, real32=False):
        """real32=True: the reference's REAL32 build of the abstract class (src/lsqr_kinds.F90:16-17): the operator's
        x and y -- and b, x, se of `solve` -- are float32 arrays (on the device: d_x, d_y point at floats)."""
        self._free()
        self._reset()
        self._user_aprod = aprod

        def trampoline(_user, mode, mm, nn, d_x, d_y, stream):
            try:
                f = self._user_aprod or self.aprod_device
                rc = f(int(mode), int(mm), int(nn), int(d_x or 0), int(d_y or 0), int(stream or 0))
                return int(rc or 0)
            except Exception as e:   # never unwind through C
                self._callback_error = e
                return 1

        self._callback_error = None
        self._cb = capi.APROD_FN(trampoline)      # keep alive as long as the handle
        h = C.c_void_p()
        create = lib().lsqrhip_create_operator_f32 if real32 else lib().lsqrhip_create_operator
        check(create(int(m), int(n), self._cb, None, C.byref(h)))
        self._h = h
        self.real32 = bool(real32)
        self.m, self.n = int(m), int(n)
        for k, v in (("atol", atol), ("btol", btol), ("conlim", conlim)):
            if v is not None:
                setattr(self, k, float(v))
        if itnlim is not None:
            self.itnlim = int(itnlim)
        if nout is not None:
            self.nout = nout
        return self

    def aprod_device(self, mode, m, n, d_x, d_y, stream):
        """mode 1: enqueue y <- y + A x ; mode 2: enqueue x <- x + A' y  (src/lsqr.f90:67-82).
        d_x (n doubles) and d_y (m doubles) are device addresses, `stream` a hipStream_t."""
        raise NotImplementedError("override aprod_device or pass aprod= to initialize")

    def lsqr(self, m, n, damp, wantse, b, atol, btol, conlim, itnlim, nout=0) -> SolveResult:
        """The reference's `lsqr` call (src/lsqr.f90:432-435) on this operator; u, v, w are
        device work vectors of the handle, x / se / the scalars come back in the result."""
        if (m, n) != (self.m, self.n):
            raise capi.LsqrHipError(capi.ERR_NOT_INIT, lib().lsqrhip_error_string(capi.ERR_NOT_INIT).decode())
        self.atol, self.btol, self.conlim, self.itnlim, self.nout = float(atol), float(btol), float(conlim), int(itnlim), nout
        return self.solve(b, damp, wantse)


class saunders_problem(lsqr_solver_ez):  # noqa: N801
    """P(m, n, nduplc, npower, damp) of the reference's test class as a device operator."""

    def __init__(self, m, n, nduplc, npower, damp, real32=False):
        """real32=True: the problem generated in binary32 arithmetic and held in real32 arrays on the device, as the
        reference's test module is under -DREAL32 (lsqrhip_lstp_create_f32)."""
        super().__init__()
        h = C.c_void_p()
        acond, rnorm = C.c_double(), C.c_double()
        create = lib().lsqrhip_lstp_create_f32 if real32 else lib().lsqrhip_lstp_create
        check(create(int(m), int(n), int(nduplc), int(npower), float(damp), C.byref(h), C.byref(acond), C.byref(rnorm)))
        self._h = h
        self.real32 = bool(real32)
        self.m, self.n = int(m), int(n)
        self.nduplc, self.npower, self.damp = int(nduplc), int(npower), float(damp)
        self.acond_lstp, self.rnorm_lstp = acond.value, rnorm.value
        self.xtrue, self.b = np.zeros(n), np.zeros(m)
        self.d, self.hy, self.hz = np.zeros(min(m, n)), np.zeros(m), np.zeros(n)
        d_b = C.c_void_p()
        check(lib().lsqrhip_lstp_vectors(h, self.xtrue.ctypes.data, self.b.ctypes.data, self.d.ctypes.data,
                                         self.hy.ctypes.data, self.hz.ctypes.data, C.byref(d_b)))
        self.d_b = d_b.value

    def test32(self) -> dict:
        """One problem of the suite under REAL32 (test/lsqrtest_module.f90:119-272 with wp = real32): acheck, LSQR with
        the tolerances the test derives from the working precision, xcheck, the error against xtrue."""
        m, n, damp = self.m, self.n, self.damp
        eps32 = float(np.finfo(np.float32).eps)
        ainform, aerr = self.acheck(eps32)                                             # :184
        self.atol = self.btol = float(np.float32(eps32) ** np.float32(0.99))          # :199-202
        self.conlim = float(np.float32(1000.0) * np.float32(self.acond_lstp))
        self.itnlim = 4 * (m + n + 50)
        self.nout = 0
        r = self.solve(self.b.astype(np.float32), damp, wantse=False)
        xinform, tests, *_ = self.xcheck(r.anorm, damp, self.b.astype(np.float32), r.x, eps32)   # :218-220
        x = r.x.astype(np.float64)
        enorm = float(np.linalg.norm(x - self.xtrue) / (1.0 + np.linalg.norm(self.xtrue)))
        return dict(m=m, n=n, nduplc=self.nduplc, npower=self.npower, damp=damp, acheck_inform=ainform, acheck_err=aerr,
                    istop=r.istop, itn=r.itn, anorm=r.anorm, acond=r.acond, rnorm=r.rnorm, arnorm=r.arnorm,
                    xnorm=r.xnorm, xcheck_inform=xinform, test1=tests[0], test2=tests[1], test3=tests[2], x=r.x,
                    enorm=enorm, success=enorm <= 0.001)

    def test(self, nout=None) -> dict:
        """One problem of the suite: test/lsqrtest_module.f90:119-272."""
        if self.real32:
            return self.test32()
        m, n, damp = self.m, self.n, self.damp
        w = nout.write if nout is not None else (lambda s: None)
        line = "-" * 34
        w("\n\n " + line + line + "\n Least-Squares Test Problem      P(" + fI(m, 5) + fI(n, 5) + fI(self.nduplc, 5)
          + fI(self.npower, 5) + fE(damp, 12, 2) + " )\n\n Condition no. =" + fE(self.acond_lstp, 12, 4)
          + "     Residual function =" + fE(self.rnorm_lstp, 17, 9) + "\n " + line + line + "\n")
        ainform, aerr = self.acheck(_EPS)                                              # :184
        w("\n\n Enter acheck.     Test of aprod for LSQR and CRAIG\n aprod seems "
          + ("OK.  " if ainform == 0 else "incorrect.") + " Relative error =" + fE(aerr, 10, 1) + "\n")
        self.atol = self.btol = _EPS ** 0.99                                           # :199-202
        self.conlim = 1000.0 * self.acond_lstp
        self.itnlim = 4 * (m + n + 50)
        self.nout = nout if nout is not None else 0
        r = self.solve(self.b, damp, wantse=False)                                     # :204-207
        xinform, tests, u, v, ww = self.xcheck(r.anorm, damp, self.b, r.x, _EPS)       # :218-220
        xn, rho1, sigma1 = np.linalg.norm(r.x), np.linalg.norm(u), np.linalg.norm(v)
        w("\n\n Enter xcheck.     Does x solve Ax = b, etc?\n    damp            =" + fE(damp, 10, 3)
          + "\n    norm(x)         =" + fE(xn, 10, 3) + "\n    norm(r)         =" + fE(rho1, 15, 8) + " = rho1"
          + "\n    norm(A'r)       =" + fE(sigma1, 10, 3) + "      = sigma1\n")
        if damp != 0.0:
            rho2 = np.sqrt(rho1 ** 2 + damp ** 2 * xn ** 2)
            w("\n    norm(s)         =" + fE(rho1 / damp, 10, 3) + "\n    norm(x,s)       =" + fE(rho2 / damp, 10, 3)
              + "\n    norm(rbar)      =" + fE(rho2, 15, 8) + " = rho2\n    norm(Abar'rbar) ="
              + fE(np.linalg.norm(ww), 10, 3) + "      = sigma2\n")
        w("\n    inform          =" + fI(xinform, 2) + "\n    tol             =" + fE(_EPS ** 0.5, 10, 3)
          + "\n    test1           =" + fE(tests[0], 10, 3) + " (Ax = b)\n    test2           =" + fE(tests[1], 10, 3)
          + " (least-squares)\n    test3           =" + fE(tests[2], 10, 3) + " (damped least-squares)\n")
        nprint = min(m, n, 8)                                                          # :224-225
        w("\n\n Solution  x:\n")
        for j0 in range(0, nprint, 4):
            w("".join(fI(j + 1, 6) + f"{r.x[j]:14.6g}" for j in range(j0, min(j0 + 4, nprint))) + "\n")
        wnorm = np.linalg.norm(r.x - self.xtrue)                                       # :233-245
        enorm = wnorm / (1.0 + np.linalg.norm(self.xtrue))
        ok = enorm <= 0.001
        w("\n LSQR  appears to " + ("be successful." if ok else "have failed.  ")
          + "     Relative error in  x  =" + fE(enorm, 10, 2) + "\n")
        return dict(m=m, n=n, nduplc=self.nduplc, npower=self.npower, damp=damp, acond_lstp=self.acond_lstp,
                    rnorm_lstp=self.rnorm_lstp, acheck_inform=ainform, acheck_err=aerr, istop=r.istop, itn=r.itn,
                    anorm=r.anorm, acond=r.acond, rnorm=r.rnorm, arnorm=r.arnorm, xnorm=r.xnorm,
                    xcheck_inform=xinform, test1=tests[0], test2=tests[1], test3=tests[2], x=r.x, enorm=enorm,
                    success=ok)


SUITE = [(m, n, 40, p, 10.0 ** (-p - 6)) for (m, n) in ((2000, 1000), (1000, 1000), (1000, 2000))
         for p in range(2, 8)]
"""lsqr_test (test/lsqrtest_module.f90:55-94): nbar = 1000, nduplc = 40, npower = ndamp = 2..7."""


def run_suite(nout=None, problems=None, real32=False) -> list[dict]:
    """The reference's 18-problem suite on the device operator; `nout` (a text stream) receives
    a log in the layout of the reference's LSQR.LIS.  real32=True: the suite of the reference's REAL32 build."""
    return [saunders_problem(*p, real32=real32).test(nout) for p in (problems or SUITE)]


if __name__ == "__main__":      # python -m lsqr_amd.operator > LSQR.LIS
    import sys
    res = run_suite(sys.stdout)
    print(f"\n {sum(r['success'] for r in res)} of {len(res)} problems successful", file=sys.stderr)
