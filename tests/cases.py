"""Parity cases shared by the golden generator and the tests.

Each case = a problem from lsqr_amd.problems + the solver options handed to
`initialize`/`solve` (reference src/lsqr.f90:91-106, 207-223).  Sizes are chosen
so the CPU checker finishes each in well under a second.
"""
from __future__ import annotations

import numpy as np

from lsqr_amd import problems as P


def _ill_conditioned():
    """Column-scaled random system: cond(A) ~ 1e8, so conlim=1e4 trips istop=4."""
    p = P.random_rows(600, 200, 6, seed=99)
    scale = np.logspace(0, -8, 200)
    a = p.a * scale[p.icol - 1]
    return P.Problem("illcond_600x200", p.m, p.n, p.irow, p.icol, a, p.b)


def _with_empty_rows_cols():
    """Rows 1..50 and columns 1..30 of a 400x250 system carry no entries."""
    p = P.random_rows(350, 220, 5, seed=5)
    return P.Problem("empty_rc_400x250", 400, 250, (p.irow + 50).astype(np.int32),
                     (p.icol + 30).astype(np.int32), p.a,
                     P.u64_to_unit(P.rng_u64(5, P.S_B, np.arange(400, dtype=np.uint64))))


def _compatible(p):
    """Replace b by A*xtrue so that Ax=b is consistent (istop=1 reachable)."""
    xtrue = 0.1 * np.arange(1, p.n + 1)
    b = np.zeros(p.m)
    np.add.at(b, p.irow - 1, p.a * xtrue[p.icol - 1])
    return P.Problem(p.name + "_compat", p.m, p.n, p.irow, p.icol, p.a, b, p.damp)


def _zero_matrix():
    return P.Problem("zero_matrix_5x4", 5, 4, np.zeros(0, np.int32), np.zeros(0, np.int32),
                     np.zeros(0), np.arange(1.0, 6.0))


def build_cases():
    """name -> (Problem, options dict)."""
    c = {}
    d = dict(damp=0.0, atol=0.0, btol=0.0, conlim=0.0, itnlim=100, wantse=False)

    def add(name, prob, **kw):
        o = dict(d)
        o.update(kw)
        c[name] = (prob, o)

    t1 = P.readme_3x3()
    add("t1_readme_default", t1)
    add("t1_readme_se", t1, wantse=True)
    add("t1_readme_damped", t1, damp=0.5, atol=1e-12, btol=1e-12)   # SURVEY 8c 'T1 damped'
    add("t2_ez_3x4", P.ez_3x4())
    add("b_zero", P.Problem("b_zero", 3, 3, t1.irow, t1.icol, t1.a, np.zeros(3)))
    add("zero_matrix", _zero_matrix())
    add("one_by_one", P.Problem("one_by_one", 1, 1, np.array([1], np.int32), np.array([1], np.int32),
                                np.array([2.5]), np.array([5.0])))
    add("poisson_20x20_it50", P.poisson2d(20, 20), itnlim=50)
    add("poisson_48x37_tol", P.poisson2d(48, 37), atol=1e-9, btol=1e-9, itnlim=2000)
    add("random_over_damped", P.random_rows(2000, 500, 8, seed=12345, damp=1e-3),
        damp=1e-3, atol=1e-8, btol=1e-8, itnlim=500)
    add("random_over_se", P.random_rows(1500, 300, 6, seed=4242), atol=1e-10, btol=1e-10,
        itnlim=500, wantse=True)
    add("random_under", P.random_rows(300, 800, 7, seed=77), atol=1e-10, btol=1e-10, conlim=1e8,
        itnlim=1000)
    add("random_compat_istop1", _compatible(P.random_rows(900, 400, 9, seed=31)),
        atol=1e-9, btol=1e-9, itnlim=1000)
    add("shuffled_dups", P.shuffled(P.random_rows(500, 60, 12, seed=3)), atol=1e-9, btol=1e-9,
        itnlim=500)
    add("powerlaw_small", P.powerlaw_rows(3000, 1200, seed=11, dmin=2, dmax=700), atol=1e-8,
        btol=1e-8, itnlim=400)
    add("illcond_conlim", _ill_conditioned(), conlim=1e4, itnlim=2000)
    add("empty_rows_cols", _with_empty_rows_cols(), atol=1e-9, btol=1e-9, itnlim=500)
    add("itnlim_1", P.random_rows(200, 100, 5, seed=8), itnlim=1)
    # Truncated twins of the four long runs whose own drift under a COO permutation (`sens`) is far above
    # 1e-10: the same systems and options stopped by itnlim where that drift is still < 1e-11 (measured with
    # the reference itself), so that they pin the kernels at the strict tolerance with identical itn, and
    # their iteration logs are compared line by line with the reference's.
    add("illcond_conlim_it10", _ill_conditioned(), conlim=1e4, itnlim=10)
    add("powerlaw_small_it10", P.powerlaw_rows(3000, 1200, seed=11, dmin=2, dmax=700), atol=1e-8,
        btol=1e-8, itnlim=10)
    add("poisson_48x37_it100", P.poisson2d(48, 37), atol=1e-9, btol=1e-9, itnlim=100)
    add("empty_rows_cols_it20", _with_empty_rows_cols(), atol=1e-9, btol=1e-9, itnlim=20)
    return c


TRUNCATED_TWINS = ["illcond_conlim_it10", "powerlaw_small_it10", "poisson_48x37_it100", "empty_rows_cols_it20"]


def assert_log_lines_match(got, want, min_records):
    """An iteration log (list of lines) against the reference's own log of the same run: text lines equal
    (5-digit scalars of the exit block may round apart), every printed iteration the same, x(1) and rnorm to
    their 10 printed digits, test1, test2, anorm, acond to their 3, phi, dknorm, dxk, alfa_opt to their 2."""
    import re

    import numpy as np
    assert len(got) == len(want)
    num = re.compile(r"[-+]?\d\.\d+E[-+]\d+")
    nrec = 0
    for a, b in zip(got, want):
        if not re.match(r"^\s+\d+\s+[-+]?\d\.\d{9}E", b):          # not an iteration record: text must be equal
            if a != b:                                                 # (exit block: 5-digit scalars may round apart)
                va, vb = [float(t) for t in num.findall(a)], [float(t) for t in num.findall(b)]
                assert num.sub("#", a) == num.sub("#", b) and np.allclose(va, vb, rtol=2e-5, atol=0)
            continue
        nrec += 1
        assert a[:6] == b[:6]                                          # the same iteration is printed
        va, vb = [float(t) for t in num.findall(a)], [float(t) for t in num.findall(b)]
        assert len(va) == len(vb)
        digits = [10, 10, 3, 3, 3, 3, 2, 2, 2, 2]
        for k, (x1, x2) in enumerate(zip(va, vb)):
            tol = 1.01 * 10.0 ** (1 - digits[k])                       # one unit of the last printed digit
            assert abs(x1 - x2) <= tol * max(abs(x2), 1e-300), (a, b, k)
    assert nrec >= min_records
