"""On-device problem generators (csrc/gen_api.h) emit exactly what lsqr_amd.problems emits."""
import numpy as np
import pytest

import oracle
from lsqr_amd import devgen, problems as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spec,host", [
    ("random:3000:700:9", lambda: P.random_rows(3000, 700, 9)),
    ("poisson2d:37:23", lambda: P.poisson2d(37, 23)),
    ("powerlaw:4000:900:1500:3", lambda: P.powerlaw_rows(4000, 900, dmin=3, dmax=1500)),
])
def test_device_generators_match_host_bit_for_bit(spec, host):
    p = host()
    irow, icol, a, b = devgen.download_coo(spec)
    assert np.array_equal(irow, p.irow) and np.array_equal(icol, p.icol) and np.array_equal(a, p.a)
    if b is not None:
        assert np.array_equal(b, p.b)
    # a row block of the same global system (what one rank of a sharded run generates)
    r0, nr = p.m // 3, p.m // 2
    irow2, icol2, a2, b2 = devgen.download_coo(spec, r0, nr)
    sel = (p.irow > r0) & (p.irow <= r0 + nr)
    assert np.array_equal(irow2, p.irow[sel] - r0) and np.array_equal(icol2, p.icol[sel])
    assert np.array_equal(a2, p.a[sel])
    if b2 is not None:
        assert np.array_equal(b2, p.b[r0:r0 + nr])


def test_solve_on_device_generated_system_matches_oracle():
    spec = "random:6000:1500:8"
    p = P.random_rows(6000, 1500, 8, damp=1e-3)
    dp = devgen.generate(spec, atol=1e-9, btol=1e-9, itnlim=300)
    assert dp.nnz == p.nnz
    from lsqr_amd.capi import DeviceBuffer
    d_x = DeviceBuffer(8 * p.n)
    r = dp.solver.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    x = d_x.to_array(np.float64, p.n)
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=1e-3, atol=1e-9, btol=1e-9, itnlim=300)
    assert (r.istop, r.itn) == (o.istop, o.itn)
    assert np.linalg.norm(x - o.x) <= 1e-10 * np.linalg.norm(o.x)
    assert abs(r.anorm - o.anorm) <= 1e-10 * o.anorm and abs(r.rnorm - o.rnorm) <= 1e-10 * o.rnorm
