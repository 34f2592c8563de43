"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes loaders for the two CPU checkers:

* ``port``  -- ``liblsqr_oracle.so``: the plain-C restatement in ``lsqr_oracle.c``
  (builds anywhere with gcc; pinned bit-for-bit by ``tests/test_oracle_golden.py``).
* ``ref``   -- ``_ref/libref_lsqr.so``: the unmodified reference compiled from
  ``/root/reference/src`` by ``oracle/Makefile`` (prebuilt; travels to the GPU box
  as a binary only).  ``ref()`` returns ``None`` when it is not there.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  Nothing under ``lsqr_amd/`` does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LOG_STRIDE = 14  # ORACLE_LOG_STRIDE

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(quiet: bool = True) -> None:
    """Compile the checker libraries (``make -C oracle``)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


@dataclass
class Result:
    """Outputs of one ``solve`` (src/lsqr.f90:207-223)."""
    x: np.ndarray
    istop: int
    itn: int
    anorm: float
    acond: float
    rnorm: float
    arnorm: float
    xnorm: float
    se: np.ndarray | None = None
    log: np.ndarray | None = field(default=None, repr=False)


def _coo(irow, icol, a):
    irow = np.ascontiguousarray(irow, dtype=np.int32)
    icol = np.ascontiguousarray(icol, dtype=np.int32)
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert irow.shape == icol.shape == a.shape
    return irow, icol, a


class _Port:
    """The C restatement (lsqr_oracle.c)."""
    kind = "port"

    def __init__(self, path: str):
        L = C.CDLL(path)
        self.L = L
        L.oracle_dnrm2.restype = C.c_double
        L.oracle_dnrm2.argtypes = [C.c_int, _f64p, C.c_int]
        L.oracle_ddot.restype = C.c_double
        L.oracle_ddot.argtypes = [C.c_int, _f64p, C.c_int, _f64p, C.c_int]
        L.oracle_dscal.restype = None
        L.oracle_dscal.argtypes = [C.c_int, C.c_double, _f64p, C.c_int]
        L.oracle_dcopy.restype = None
        L.oracle_dcopy.argtypes = [C.c_int, _f64p, C.c_int, _f64p, C.c_int]
        L.oracle_d2norm.restype = C.c_double
        L.oracle_d2norm.argtypes = [C.c_double, C.c_double]
        L.oracle_aprod.restype = C.c_int
        L.oracle_aprod.argtypes = [C.c_int, C.c_int, C.c_int, C.c_longlong, _i32p, _i32p, _f64p,
                                   _f64p, _f64p, _f64p]
        L.oracle_validate.restype = C.c_int
        L.oracle_validate.argtypes = [C.c_int, C.c_int, C.c_longlong, _i32p, _i32p]
        L.oracle_lsqr_ez.restype = C.c_int
        L.oracle_lsqr_ez.argtypes = [C.c_int, C.c_int, C.c_longlong, _i32p, _i32p, _f64p, _f64p,
                                     C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                     C.c_int, _f64p, _f64p] + [C.c_void_p] * 7 + [C.c_void_p, C.c_int]
        L.oracle_acheck.restype = C.c_int
        L.oracle_acheck.argtypes = [C.c_int, C.c_int, C.c_longlong, _i32p, _i32p, _f64p, C.c_double,
                                    C.c_void_p]
        L.oracle_xcheck.restype = C.c_int
        L.oracle_xcheck.argtypes = [C.c_int, C.c_int, C.c_longlong, _i32p, _i32p, _f64p, C.c_double,
                                    C.c_double, C.c_double, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p]

        L.oracle_lstp_generate.restype = C.c_int
        L.oracle_lstp_generate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _f64p, _f64p, _f64p,
                                           _f64p, _f64p, C.c_void_p, C.c_void_p]
        L.oracle_lstp_test.restype = C.c_int
        L.oracle_lstp_test.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _f64p, _f64p, _f64p, _f64p]

    # the reference's test-problem class (lstp_oracle.c) -------------------
    def lstp_generate(self, m, n, nduplc, npower, damp):
        """lstp (test/lsqrtest_module.f90:422-505): dict(xtrue, b, d, hy, hz, acond, rnorm)."""
        xtrue, b, d, hy, hz = np.zeros(n), np.zeros(m), np.zeros(min(m, n)), np.zeros(m), np.zeros(n)
        acond, rnorm = C.c_double(), C.c_double()
        rc = self.L.oracle_lstp_generate(m, n, nduplc, npower, damp, xtrue, b, d, hy, hz,
                                         C.addressof(acond), C.addressof(rnorm))
        assert rc == 0
        return dict(xtrue=xtrue, b=b, d=d, hy=hy, hz=hz, acond=acond.value, rnorm=rnorm.value)

    def lstp_test(self, m, n, nduplc, npower, damp):
        """One problem of the 18-problem suite (test/lsqrtest_module.f90:119-272)."""
        x, xtrue, b, res = np.zeros(n), np.zeros(n), np.zeros(m), np.zeros(16)
        rc = self.L.oracle_lstp_test(m, n, nduplc, npower, damp, x, xtrue, b, res)
        assert rc == 0
        keys = ("acond_lstp", "rnorm_lstp", "acheck_inform", "acheck_err", "istop", "itn", "anorm", "acond",
                "rnorm", "arnorm", "xnorm", "xcheck_inform", "test1", "test2", "test3", "enorm")
        out = dict(zip(keys, res.tolist()))
        for k in ("acheck_inform", "istop", "itn", "xcheck_inform"):
            out[k] = int(out[k])
        out.update(x=x, xtrue=xtrue, b=b)
        return out

    # BLAS-1 -------------------------------------------------------------
    def dnrm2(self, x, incx=1, n=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = (len(x) + incx - 1) // incx if n is None else n
        return float(self.L.oracle_dnrm2(n, x, incx))

    def ddot(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        return float(self.L.oracle_ddot(len(x), x, 1, y, 1))

    def dscal(self, da, x):
        x = np.array(x, dtype=np.float64)
        self.L.oracle_dscal(len(x), float(da), x, 1)
        return x

    def dcopy(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self.L.oracle_dcopy(len(x), x, 1, y, 1)
        return y

    def d2norm(self, a, b):
        return float(self.L.oracle_d2norm(float(a), float(b)))

    # test hook ----------------------------------------------------------
    def set_norm_ulp(self, itn: int, which: int, ulps: int) -> None:
        """Move beta (which=1) or alpha (which=2) of iteration `itn` (0 = the start) of every following solve by
        `ulps` units in the last place; which=0 switches it off (lsqr_oracle.c oracle_set_norm_ulp)."""
        self.L.oracle_set_norm_ulp.restype = None
        self.L.oracle_set_norm_ulp.argtypes = [C.c_int, C.c_int, C.c_int]
        self.L.oracle_set_norm_ulp(int(itn), int(which), int(ulps))

    def set_accurate_sums(self, on: bool) -> None:
        """Compensated row sums in aprod and pairwise sums of squares in dnrm2 for every following call (test hooks
        oracle_set_accurate_rowsums / oracle_set_norm_order): the same recurrences with correctly rounded sums."""
        self.L.oracle_set_accurate_rowsums.restype = None
        self.L.oracle_set_accurate_rowsums.argtypes = [C.c_int]
        self.L.oracle_set_norm_order.restype = None
        self.L.oracle_set_norm_order.argtypes = [C.c_int]
        self.L.oracle_set_accurate_rowsums(1 if on else 0)
        self.L.oracle_set_norm_order(1 if on else 0)

    # operator -----------------------------------------------------------
    def validate(self, m, n, irow, icol):
        irow = np.ascontiguousarray(irow, dtype=np.int32)
        icol = np.ascontiguousarray(icol, dtype=np.int32)
        return int(self.L.oracle_validate(m, n, len(irow), irow, icol))

    def aprod(self, mode, m, n, irow, icol, a, x, y):
        """Returns the updated (x, y) copies: mode 1 -> y += A x, mode 2 -> x += A' y."""
        irow, icol, a = _coo(irow, icol, a)
        x = np.array(x, dtype=np.float64)
        y = np.array(y, dtype=np.float64)
        scratch = np.empty(max(m, n, 1))
        rc = self.L.oracle_aprod(mode, m, n, len(a), irow, icol, a, x, y, scratch)
        if rc:
            raise ValueError("invalid mode in aprod_ez")
        return x, y

    def solve(self, m, n, irow, icol, a, b, damp=0.0, atol=0.0, btol=0.0, conlim=0.0,
              itnlim=100, wantse=False, want_log=False) -> Result:
        irow, icol, a = _coo(irow, icol, a)
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert b.shape == (m,)
        x = np.zeros(max(n, 1))
        se = np.zeros(max(n, 1))
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        log = np.zeros((itnlim, LOG_STRIDE)) if want_log else None
        rc = self.L.oracle_lsqr_ez(m, n, len(a), irow, icol, a, b, damp, atol, btol, conlim, itnlim,
                                   int(wantse), x, se, C.addressof(istop), C.addressof(itn),
                                   *[C.addressof(s) for s in sc],
                                   log.ctypes.data if want_log else None, itnlim if want_log else 0)
        if rc:
            raise MemoryError("oracle_lsqr_ez")
        return Result(x[:n], istop.value, itn.value, *[s.value for s in sc],
                      se=se[:n] if wantse else None,
                      log=log[:itn.value] if want_log else None)

    def acheck(self, m, n, irow, icol, a, eps=np.finfo(np.float64).eps):
        irow, icol, a = _coo(irow, icol, a)
        err = C.c_double()
        inform = self.L.oracle_acheck(m, n, len(a), irow, icol, a, eps, C.addressof(err))
        return int(inform), err.value

    def xcheck(self, m, n, irow, icol, a, anorm, damp, b, x, eps=np.finfo(np.float64).eps):
        irow, icol, a = _coo(irow, icol, a)
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        u, v, w, tests = np.zeros(m), np.zeros(n), np.zeros(n), np.zeros(3)
        inform = self.L.oracle_xcheck(m, n, len(a), irow, icol, a, anorm, damp, eps, b, x, u, v, w, tests)
        return int(inform), tests, u, v, w


class _Ref:
    """The unmodified reference behind oracle/ref_shim.f90."""
    kind = "reference"

    def __init__(self, path: str):
        L = C.CDLL(path)
        self.L = L
        L.ref_dnrm2.restype = C.c_double
        L.ref_dnrm2.argtypes = [C.c_int, _f64p, C.c_int]
        L.ref_ddot.restype = C.c_double
        L.ref_ddot.argtypes = [C.c_int, _f64p, C.c_int, _f64p, C.c_int]
        L.ref_dscal.restype = None
        L.ref_dscal.argtypes = [C.c_int, C.c_double, _f64p, C.c_int]
        L.ref_dcopy.restype = None
        L.ref_dcopy.argtypes = [C.c_int, _f64p, C.c_int, _f64p, C.c_int]
        L.ref_aprod.restype = None
        L.ref_aprod.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p]
        L.ref_lsqr_ez.restype = None
        L.ref_lsqr_ez.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p, C.c_double,
                                  C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _f64p, _f64p] + \
                                 [C.c_void_p] * 7 + [C.c_char_p, C.c_int]
        L.ref_acheck.restype = None
        L.ref_acheck.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_double, C.c_void_p]
        L.ref_xcheck.restype = None
        L.ref_xcheck.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_double, C.c_double,
                                 C.c_double, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_void_p, _f64p]

    def dnrm2(self, x, incx=1, n=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = (len(x) + incx - 1) // incx if n is None else n
        return float(self.L.ref_dnrm2(n, x, incx))

    def ddot(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        return float(self.L.ref_ddot(len(x), x, 1, y, 1))

    def dscal(self, da, x):
        x = np.array(x, dtype=np.float64)
        self.L.ref_dscal(len(x), float(da), x, 1)
        return x

    def dcopy(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self.L.ref_dcopy(len(x), x, 1, y, 1)
        return y

    def aprod(self, mode, m, n, irow, icol, a, x, y):
        irow, icol, a = _coo(irow, icol, a)
        x = np.array(x, dtype=np.float64)
        y = np.array(y, dtype=np.float64)
        self.L.ref_aprod(mode, m, n, len(a), irow, icol, a, x, y)
        return x, y

    def solve(self, m, n, irow, icol, a, b, damp=0.0, atol=0.0, btol=0.0, conlim=0.0,
              itnlim=100, wantse=False, logpath: str | None = None) -> Result:
        irow, icol, a = _coo(irow, icol, a)
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(max(n, 1))
        se = np.zeros(max(n, 1))
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        lp = logpath.encode() if logpath else b""
        self.L.ref_lsqr_ez(m, n, len(a), irow, icol, a, b, damp, atol, btol, conlim, itnlim,
                           int(wantse), x, se, C.addressof(istop), C.addressof(itn),
                           *[C.addressof(s) for s in sc], lp, len(lp))
        return Result(x[:n], istop.value, itn.value, *[s.value for s in sc],
                      se=se[:n] if wantse else None)

    def acheck(self, m, n, irow, icol, a, eps=np.finfo(np.float64).eps):
        irow, icol, a = _coo(irow, icol, a)
        inform = C.c_int()
        self.L.ref_acheck(m, n, len(a), irow, icol, a, eps, C.addressof(inform))
        return inform.value

    def xcheck(self, m, n, irow, icol, a, anorm, damp, b, x, eps=np.finfo(np.float64).eps):
        irow, icol, a = _coo(irow, icol, a)
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        u, v, w, tests = np.zeros(m), np.zeros(n), np.zeros(n), np.zeros(3)
        inform = C.c_int()
        self.L.ref_xcheck(m, n, len(a), irow, icol, a, anorm, damp, eps, b, x, u, v, w,
                          C.addressof(inform), tests)
        return inform.value, tests, u, v, w


_port = None
_ref = None


SUITE = [(m, n, 40, p, 10.0 ** (-p - 6)) for (m, n) in ((2000, 1000), (1000, 1000), (1000, 2000))
         for p in range(2, 8)]
"""The 18 problems of lsqr_test (test/lsqrtest_module.f90:55-94): (m, n, nduplc, npower, damp)."""


def parse_lis(text: str) -> list[dict]:
    """Numeric fields of an LSQR.LIS log (the file the reference's test program writes), one
    dict per problem: header, acheck error, exit scalars, xcheck inform/tests, x(1:8), verdict."""
    import re
    num = r"[-+]?\d*\.?\d+(?:[EeDd][-+]?\d+)?"
    out = []
    for blk in text.split("Least-Squares Test Problem")[1:]:
        d = {}
        h = re.search(r"P\(\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(" + num + r")\s*\)", blk)
        d["m"], d["n"], d["nduplc"], d["npower"] = (int(h.group(i)) for i in range(1, 5))
        d["damp"] = float(h.group(5))
        d["acond_lstp"] = float(re.search(r"Condition no\. =\s*(" + num + ")", blk).group(1))
        d["rnorm_lstp"] = float(re.search(r"Residual function =\s*(" + num + ")", blk).group(1))
        d["acheck_err"] = float(re.search(r"aprod seems OK\.\s+Relative error =\s*(" + num + ")", blk).group(1))
        d["istop"] = int(re.search(r"istop\s+=\s*(\d+)", blk).group(1))
        d["itn"] = int(re.search(r"itn\s+=\s*(\d+)", blk).group(1))
        for k in ("anorm", "acond", "bnorm", "xnorm", "rnorm", "arnorm"):
            d[k] = float(re.search(r"Exit  LSQR\..*?\b" + k + r"\s*=\s*(" + num + ")", blk).group(1))
        d["xcheck_inform"] = int(re.search(r"inform\s+=\s*(\d+)", blk).group(1))
        for i in (1, 2, 3):
            d[f"test{i}"] = float(re.search(rf"test{i}\s+=\s*(" + num + ")", blk).group(1))
        sol = blk.split("Solution  x:")[1].split("LSQR  appears")[0]
        d["x8"] = [float(v) for _, v in re.findall(r"(\d+)\s+(" + num + ")", sol)][:8]
        d["success"] = "appears to be successful" in blk
        d["enorm"] = float(re.search(r"Relative error in  x  =\s*(" + num + ")", blk).group(1))
        out.append(d)
    return out


def port() -> _Port:
    """The C restatement; built on demand (gcc only)."""
    global _port
    if _port is None:
        path = os.path.join(_HERE, "liblsqr_oracle.so")
        src = os.path.join(_HERE, "lsqr_oracle.c")
        san = os.environ.get("LSQR_ORACLE_LIB")    # tests/test_oracle_sanitized.py: the ASan / UBSan build of the same sources
        if san:
            path = san
        elif not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _port = _Port(path)
    return _port


def ref() -> _Ref | None:
    """The compiled reference, or None when oracle/_ref was not built/shipped."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libref_lsqr.so")
        if not os.path.exists(path):
            return None
        _ref = _Ref(path)
    return _ref
