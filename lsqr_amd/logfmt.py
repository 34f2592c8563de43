"""Iteration log of `nout /= 0`, formatted from device records.

Reproduces, character for character, what reference src/lsqr.f90 writes:
header :589-595, column titles + itn-0 line :655-671, per-iteration lines with the
selective print rule :813-837, exit block :872-880.  The device only hands over one
record of doubles per iteration (include/lsqrhip.h, LSQRHIP_LOG_STRIDE); Fortran
edit descriptors are emulated here.
"""
from __future__ import annotations

import math

ENTER = " Enter LSQR.  "
EXIT = " Exit  LSQR.  "
MSG = [
    "The exact solution is x = 0                          ",
    "A solution to Ax = b was found, given atol, btol     ",
    "A least-squares solution was found, given atol       ",
    "A damped least-squares solution was found, given atol",
    "Cond(Abar) seems to be too large, given conlim       ",
    "The iteration limit was reached                      ",
]


def fE(v: float, w: int, d: int) -> str:
    """Fortran `1P,Ew.d`: one digit before the point, two-digit exponent ('E' dropped for three)."""
    if math.isnan(v):
        return "NaN".rjust(w)
    if math.isinf(v):
        return ("Inf" if v > 0 else "-Inf").rjust(w)
    s = f"{v:.{d}E}"
    mant, exp = s.split("E")
    e = int(exp)
    if abs(e) >= 100:
        s = f"{mant}{'+' if e >= 0 else '-'}{abs(e):03d}"
    else:
        s = f"{mant}E{'+' if e >= 0 else '-'}{abs(e):02d}"
    return s.rjust(w) if len(s) <= w else "*" * w


def fI(v: int, w: int) -> str:
    s = str(int(v))
    return s.rjust(w) if len(s) <= w else "*" * w


def iter_line(rec) -> str:
    # '(1P, I6, 2E17.9, 4E10.2, E9.1, 3E8.1)'   src/lsqr.f90:828-829
    return (fI(rec[0], 6) + fE(rec[1], 17, 9) + fE(rec[2], 17, 9) + "".join(fE(rec[k], 10, 2) for k in (3, 4, 5, 6))
            + fE(rec[7], 9, 1) + "".join(fE(rec[k], 8, 1) for k in (8, 9, 10)))


def format_log(m, n, damp, wantse, atol, btol, conlim, itnlim, records, result, bnorm, dxmax, maxdx,
               test2_0, beta0) -> str:
    out = []
    w = out.append
    # :590-594
    w("\n\n" + ENTER + "     Least-squares solution of  Ax = b\n")
    w(" The matrix  A  has" + fI(m, 7) + " rows   and" + fI(n, 7) + " columns\n")
    w(" damp   =" + fE(damp, 22, 14) + "   wantse =" + ("T" if wantse else "F").rjust(10) + "\n")
    w(" atol   =" + fE(atol, 10, 2) + " " * 15 + "conlim =" + fE(conlim, 10, 2) + "\n")
    w(" btol   =" + fE(btol, 10, 2) + " " * 15 + "itnlim =" + fI(itnlim, 10) + "\n")
    damped = damp > 0.0
    ctol = 1.0 / conlim if conlim > 0.0 else 0.0
    loop_ran = result.itn > 0 or result.istop != 0
    if loop_ran:
        # :655-670
        if damped:
            w("\n\n   Itn       x(1)           Function     Compatible   LS     Norm Abar Cond Abar\n")
        else:
            w("\n\n   Itn       x(1)           Function     Compatible   LS        Norm A    Cond A\n")
        w(" " * 80 + "    phi    dknorm   dxk  alfa_opt\n")
        w(fI(0, 6) + fE(0.0, 17, 9) + fE(beta0, 17, 9) + fE(1.0, 10, 2) + fE(test2_0, 10, 2) + "\n")
        w("\n")
        for rec in records:
            itn = int(rec[0])
            test1, test2, acond, istop, rtol = rec[3], rec[4], rec[6], int(rec[11]), rec[12]
            test3 = 1.0 / acond if acond != 0.0 else math.inf
            show = (n <= 40 or itn <= 10 or itn >= itnlim - 10 or itn % 10 == 0 or test3 <= 2.0 * ctol
                    or test2 <= 10.0 * atol or test1 <= 10.0 * rtol or istop != 0)      # :815-822
            if show:
                w(iter_line(rec) + "\n")
    # :873-879
    w("\n\n" + EXIT + " " * 5 + "istop  =" + fI(result.istop, 2) + " " * 15 + "itn    =" + fI(result.itn, 8) + "\n")
    w(EXIT + " " * 5 + "anorm  =" + fE(result.anorm, 12, 5) + " " * 5 + "acond  =" + fE(result.acond, 12, 5) + "\n")
    w(EXIT + " " * 5 + "bnorm  =" + fE(bnorm, 12, 5) + " " * 5 + "xnorm  =" + fE(result.xnorm, 12, 5) + "\n")
    w(EXIT + " " * 5 + "rnorm  =" + fE(result.rnorm, 12, 5) + " " * 5 + "arnorm =" + fE(result.arnorm, 12, 5) + "\n")
    w(EXIT + " " * 5 + "max dx =" + fE(dxmax, 8, 1) + " occurred at itn " + fI(maxdx, 8) + "\n")
    w(EXIT + " " * 5 + "       =" + fE(dxmax / (result.xnorm + 1.0e-20), 8, 1) + "*xnorm\n")
    w(EXIT + " " * 5 + MSG[result.istop] + "\n")
    return "".join(out)
