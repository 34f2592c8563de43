"""Range safety of the norms inside the loop (reference dnrm2, src/lsqrblas.f90:123-159, is the
scaled dlassq recurrence: no over- or underflow anywhere in the fp64 range).  The same systems
with b and A pushed to 1e+-200 / 1e+-160 must give the reference's istop, itn, x, anorm, rnorm
through every launch schedule and every layout."""
import os

import numpy as np
import pytest

import oracle
from lsqr_amd import problems as P
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu

SCALES = [("b*1e200", 1.0, 1e200), ("b*1e-200", 1.0, 1e-200), ("A*1e160", 1e160, 1.0), ("A*1e-160", 1e-160, 1.0),
          ("A*1e160,b*1e-200", 1e160, 1e-200), ("A*1e-150,b*1e150", 1e-150, 1e150)]


def xerr(x, xo):
    """relative 2-norm error, formed without squaring 1e+-200"""
    sc = np.max(np.abs(xo))
    if sc == 0.0:                       # x itself underflowed in the reference (A*1e160, b*1e-200): so must ours
        return 0.0 if not np.any(x) else np.inf
    return np.linalg.norm((x - xo) / sc) / np.linalg.norm(xo / sc)


def run_case(p, sa, sb, damp, kw, layouts):
    a, b = p.a * sa, p.b * sb
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, a, b, damp=damp, **kw)
    assert o.itn >= 1
    for env in layouts:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            s = lsqr_solver_ez().initialize(p.m, p.n, a, p.irow, p.icol, **kw)
            for pipeline in (0, 1, 2):
                s.set_option("pipeline", pipeline)
                r = s.solve(b, damp)
                assert (r.istop, r.itn) == (o.istop, o.itn), (env, pipeline, r.istop, r.itn, o.istop, o.itn)
                assert xerr(r.x, o.x) <= 1e-10, (env, pipeline)
                assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm, (env, pipeline, r.anorm, o.anorm)
                assert abs(r.rnorm - o.rnorm) <= 1e-10 * o.rnorm, (env, pipeline, r.rnorm, o.rnorm)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v


@pytest.mark.parametrize("label,sa,sb", SCALES)
def test_scaled_random_system_all_schedules_and_layouts(label, sa, sb):
    p = P.random_rows(2000, 500, 8, seed=12345)
    # damp scales with A so that the damped problem is the same problem at every scale
    run_case(p, sa, sb, 1e-3 * sa, dict(atol=1e-8, btol=1e-8, itnlim=500),
             [{}, {"LSQRHIP_CSB": "1", "LSQRHIP_CSB_R": "300"},
              {"LSQRHIP_PANELS": "1", "LSQRHIP_PANEL_KB": "64", "LSQRHIP_XLDS": "0", "LSQRHIP_CSB": "0"},
              {"LSQRHIP_XLDS": "1", "LSQRHIP_XLDS_COLS": "1024"}])


@pytest.mark.parametrize("label,sa,sb", SCALES)
def test_scaled_poisson_sliced_ell(label, sa, sb):
    p = P.poisson2d(40, 30)
    run_case(p, sa, sb, 0.0, dict(itnlim=40), [{}, {"LSQRHIP_PAT": "0"}, {"LSQRHIP_PAT": "0", "LSQRHIP_SELLP": "0"},
                                                {"LSQRHIP_PAT": "0", "LSQRHIP_VAL8": "0", "LSQRHIP_SPAT": "1"},
                                                {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_SELL": "0"}])


def test_norm_of_b_with_elements_across_the_whole_range():
    """Blue's accumulators: huge, tiny and ordinary elements in one b."""
    p = P.random_rows(600, 200, 6, seed=3)
    po = oracle.port()
    for pattern in ([1e200, 1e-200, 1.0], [3e153, 2e153, 1.0], [1e-170, 3e-170, 0.0], [1e300, 1e300, 1e300]):
        b = p.b.copy()
        b[:3] = pattern
        if pattern[2] == 0.0:
            b[3:] = 0.0
        o = po.solve(p.m, p.n, p.irow, p.icol, p.a, b, itnlim=8)
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=8)
        r = s.solve(b, 0.0)
        assert (r.istop, r.itn) == (o.istop, o.itn)
        assert xerr(r.x, o.x) <= 1e-10
        assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * o.rnorm


def test_operator_handle_norms_are_range_safe():
    """A user device operator has no matrix to take a scale from: its norms use Blue's form."""
    from lsqr_amd.capi import check, lib
    from lsqr_amd.operator import lsqr_solver_device

    class Wrapped(lsqr_solver_device):
        def __init__(self, ez):
            super().__init__()
            self.ez = ez

        def aprod_device(self, mode, m, n, d_x, d_y, stream):
            check(lib().lsqrhip_set_stream(self.ez._h, stream))
            check(lib().lsqrhip_aprod_device(self.ez._h, mode, d_x, d_y))
            return 0

    p = P.random_rows(800, 300, 7, seed=5)
    for sa, sb in ((1e160, 1e-200), (1e-160, 1.0), (1.0, 1e200)):
        a, b = p.a * sa, p.b * sb
        o = oracle.port().solve(p.m, p.n, p.irow, p.icol, a, b, atol=1e-9, btol=1e-9, itnlim=300)
        ez = lsqr_solver_ez().initialize(p.m, p.n, a, p.irow, p.icol)
        op = Wrapped(ez).initialize(p.m, p.n, atol=1e-9, btol=1e-9, itnlim=300)
        r = op.solve(b, 0.0)
        assert (r.istop, r.itn) == (o.istop, o.itn)
        assert xerr(r.x, o.x) <= 1e-10
        assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * o.rnorm


@pytest.mark.parametrize("layout", [{}, {"LSQRHIP_CSB": "1"}])
def test_changing_the_norm_exponent_rebuilds_the_captured_batches(layout):
    """The power-of-two scale of the fused norms travels BY VALUE in every captured kernel node, while the
    scalar steps read its inverse from the state of the solve: a handle that has solved through its graphs
    and is then given another exponent (the ranks of a sharded solve agree on one, `norm_exp`) must capture
    them again -- otherwise beta and alpha come out wrong by 2^(e' - e) without any error."""
    p = P.random_rows(3000, 800, 8, seed=4, damp=1e-3)
    kw = dict(atol=1e-9, btol=1e-9, itnlim=200)
    old = {k: os.environ.get(k) for k in layout}
    os.environ.update(layout)
    try:
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, **kw)
        r0 = s.solve(p.b, p.damp)                                  # captures the batches under the build's exponent
        e = s.get_option("norm_exp")
        for de in (3, -3, 40):
            s.set_option("norm_exp", e + de)
            r = s.solve(p.b, p.damp)
            f = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, **kw)
            f.set_option("norm_exp", e + de)                       # a fresh handle: captured under the new exponent
            g = f.solve(p.b, p.damp)
            assert (r.istop, r.itn) == (g.istop, g.itn) == (r0.istop, r0.itn)
            assert np.array_equal(r.x, g.x) and (r.anorm, r.rnorm, r.xnorm) == (g.anorm, g.rnorm, g.xnorm)
            assert xerr(r.x, r0.x) <= 1e-12                         # (a power-of-two scale: the same norms to rounding)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
