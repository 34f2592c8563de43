"""On-device problem generators (csrc/gen_api.h) emit exactly what lsqr_amd.problems emits."""
import os

import numpy as np
import pytest

import oracle
from lsqr_amd import devgen, problems as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spec,host", [
    ("random:3000:700:9", lambda: P.random_rows(3000, 700, 9)),
    ("poisson2d:37:23", lambda: P.poisson2d(37, 23)),
    ("mesh2d:61:47:5:3", lambda: P.mesh2d(61, 47, 5, 3)),
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


def test_scale_properties_on_100M_nonzero_random_system():
    """config 3 shape at 1e8 nonzeros (4M x 1M, 25 per row, generated in HBM; column panels
    are chosen automatically).  (Oracle parity at this scale: tests/test_gpu_fullsize_parity.py.)  Here the
    size-independent checks: the reference's own acheck (mode 1 vs mode 2 consistency) and
    xcheck (does x solve the damped problem?) on the device operator, linearity, determinism."""
    from lsqr_amd.capi import DeviceBuffer
    spec = "random:4000000:1000000:25"
    dp = devgen.generate(spec, atol=1e-10, btol=1e-10, itnlim=400)
    s = dp.solver
    assert dp.nnz == 100_000_000
    inform, err = s.acheck()
    assert inform == 0 and err < 1e-12
    d_x = DeviceBuffer(8 * dp.n)
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    x = d_x.to_array(np.float64, dp.n)
    b = dp.d_b.to_array(np.float64, dp.m)
    assert r.istop == 3 and 10 < r.itn < 400               # damped least squares, converged on atol
    inform, tests, u, v, w = s.xcheck(r.anorm, 1e-3, b, x)
    assert inform in (1, 2, 3) and tests[2] < 1e-7          # A'r - damp^2 x ~ 0
    r2 = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    assert (r2.itn, r2.anorm, r2.rnorm) == (r.itn, r.anorm, r.rnorm)
    assert np.array_equal(d_x.to_array(np.float64, dp.n), x)
    # linearity of aprod on the panelled operator
    xa = P.u64_to_unit(P.rng_u64(1, 9, np.arange(dp.n, dtype=np.uint64)))
    ya, y2 = np.zeros(dp.m), np.zeros(dp.m)
    s.aprod(1, dp.m, dp.n, xa, ya)
    s.aprod(1, dp.m, dp.n, 3.0 * xa, y2)
    assert np.max(np.abs(y2 - 3.0 * ya)) <= 1e-13 * np.max(np.abs(y2))


def test_scale_properties_beyond_2_to_31_nonzeros_with_lds_panels():
    """4M x 1M with 600 per row = 2.4e9 nonzeros (> 2^31: 64-bit row pointers at real scale; LDS
    column panels chosen automatically): BASELINE config 3's shape at 60 % of its literal size, so
    that the test needs ~140 GB of HBM and a few seconds.  Size-independent checks: acheck's adjoint
    identity, xcheck on the solution of the damped problem, determinism."""
    import torch
    from lsqr_amd.capi import DeviceBuffer
    free, total = torch.cuda.mem_get_info()
    if free < 180e9:
        pytest.skip("needs ~140 GB of free HBM")
    spec = "random:4000000:1000000:600"
    old = os.environ.get("LSQRHIP_CSB")
    os.environ["LSQRHIP_CSB"] = "2"     # round 2's first rule: the LDS panels keep this matrix (csb_rule would sweep it)
    try:
        dp = devgen.generate(spec, atol=1e-9, btol=1e-9, itnlim=60)
    finally:
        os.environ.pop("LSQRHIP_CSB", None)
        if old is not None:
            os.environ["LSQRHIP_CSB"] = old
    s = dp.solver
    info = s.info()
    assert dp.nnz == 2_400_000_000 and info["rowptr_bytes"] == 8
    assert info["xlds"] == 2 and info["xlds_t"] == 2
    assert info["col_bytes"] == (4 if os.environ.get("LSQRHIP_COL16") == "0" else 2)   # the ablation knob, if set
    inform, err = s.acheck()
    assert inform == 0 and err < 1e-12
    d_x = DeviceBuffer(8 * dp.n)
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    x = d_x.to_array(np.float64, dp.n)
    b = dp.d_b.to_array(np.float64, dp.m)
    assert r.istop == 3 and 3 < r.itn < 60
    inform, tests, u, v, w = s.xcheck(r.anorm, 1e-3, b, x)
    assert inform in (1, 2, 3) and tests[2] < 1e-7
    r2 = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    assert (r2.itn, r2.anorm, r2.rnorm) == (r.itn, r.anorm, r.rnorm)
    assert np.array_equal(d_x.to_array(np.float64, dp.n), x)


def test_scale_properties_of_baseline_config3_at_its_literal_size():
    """BASELINE configs[2] LITERALLY: 4M x 1M with 0.1 % nonzeros = 1000 per row = 4.0e9 nonzeros (just
    under the 2^32 this build's sort can index; column-swept row blocks chosen automatically: 8.5 ms per
    product against the LDS column panels' 10.1).  Needs ~210 GB of HBM while it builds.  Size-independent checks: acheck's adjoint
    identity, a short solve that converges on atol (damped least squares), repeats itself bit for bit and
    passes the reference's xcheck."""
    import torch
    from lsqr_amd.capi import DeviceBuffer
    free, total = torch.cuda.mem_get_info()
    if free < 230e9:
        pytest.skip("needs ~210 GB of free HBM")
    dp = devgen.generate("random:4000000:1000000:1000", atol=1e-8, btol=1e-8, itnlim=40)
    # (round 4: the build composes the fill's 4-byte permutation and releases its sort buffers before the layout is
    # allocated, and never keeps them beside two finished layouts -- the peak beyond the 64 GB of triplets: 113 GB for
    # a 96 GB result, 177 GB in all; round 3: ~210 GB)
    peak, kept = dp.solver.get_option("build_peak_bytes"), dp.solver.get_option("build_kept_bytes")
    print(f"build peak {peak / 1e9:.1f} GB beyond the 64 GB of triplets, kept {kept / 1e9:.1f} GB")
    assert kept < 100e9 and peak < 125e9
    s = dp.solver
    info = s.info()
    assert dp.nnz == 4_000_000_000
    assert info["xlds"] == 3 and info["xlds_t"] == 3
    inform, err = s.acheck()
    assert inform == 0 and err < 1e-12
    d_x = DeviceBuffer(8 * dp.n)
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    x = d_x.to_array(np.float64, dp.n)
    b = dp.d_b.to_array(np.float64, dp.m)
    assert r.istop == 3 and 3 < r.itn < 40
    inform, tests, u, v, w = s.xcheck(r.anorm, 1e-3, b, x)
    assert inform in (1, 2, 3) and tests[2] < 1e-6
    r2 = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
    assert (r2.itn, r2.anorm, r2.rnorm) == (r.itn, r.anorm, r.rnorm)
    assert np.array_equal(d_x.to_array(np.float64, dp.n), x)


@pytest.mark.parametrize("spec,expect", [
    ("random:10000000:10000000:100", dict(nnz=1_000_000_000, panels=39)),      # BASELINE configs[3]: what --gpus N shards
    ("powerlaw:5000000:2000000:10000", dict(panels=8)),                        # BASELINE configs[4]: skewed rows
])
def test_scale_properties_of_the_panelled_baseline_configurations(spec, expect):
    """BASELINE configs[3] and configs[4] at their full size, generated in HBM: column-swept row blocks
    (csrc/csb.h; with LSQRHIP_CSB=0 the L2 column panels with the grid chosen by timing, and for the
    power law the long-segment waves of spmv.h phase 2b).  Oracle parity of these shapes -- configs[4] whole,
    one rank's block of configs[3] -- is held by tests/test_gpu_fullsize_parity.py; the whole 1e9-nonzero
    configs[3] (16 GB of triplets, ~10 s per oracle iteration) gets the size-independent checks here: acheck's adjoint identity (A and A' are built and laid out
    independently), linearity, bit-level determinism of both products (neither the tuned grid nor
    which wave takes which long segment may show), and a short solve that repeats itself exactly
    and agrees across the launch schedules."""
    import torch
    from lsqr_amd.capi import DeviceBuffer
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~80 GB of free HBM")
    dp = devgen.generate(spec, itnlim=12)
    s = dp.solver
    info = s.info()
    if "nnz" in expect:
        assert dp.nnz == expect["nnz"]
    if os.environ.get("LSQRHIP_CSB") == "0":      # ablation knob: the L2 column panels of round 1
        assert info["panels"] == expect["panels"] and info["xlds"] == 0 and info["sell"] == 0
    else:                                          # column-swept row blocks (csrc/csb.h) for A and A'
        assert info["xlds"] == 3 and info["xlds_t"] == 3 and info["panels"] == 1 and info["sell"] == 0
    inform, err = s.acheck()
    assert inform == 0 and err < 1e-12
    xa = P.u64_to_unit(P.rng_u64(1, 9, np.arange(dp.n, dtype=np.uint64)))
    yb = P.u64_to_unit(P.rng_u64(2, 9, np.arange(dp.m, dtype=np.uint64)))
    y1, y2, y3 = np.zeros(dp.m), np.zeros(dp.m), np.zeros(dp.m)
    s.aprod(1, dp.m, dp.n, xa, y1)
    s.aprod(1, dp.m, dp.n, xa, y2)
    s.aprod(1, dp.m, dp.n, 3.0 * xa, y3)
    assert np.array_equal(y1, y2)
    assert np.max(np.abs(y3 - 3.0 * y1)) <= 1e-13 * np.max(np.abs(y3))
    x1, x2 = np.zeros(dp.n), np.zeros(dp.n)
    s.aprod(2, dp.m, dp.n, x1, yb)
    s.aprod(2, dp.m, dp.n, x2, yb)
    assert np.array_equal(x1, x2)
    d_x = DeviceBuffer(8 * dp.n)
    out = []
    for pipeline in (2, 2, 1):
        s.set_option("pipeline", pipeline)
        r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
        out.append((r.istop, r.itn, r.anorm, r.rnorm, r.xnorm, d_x.to_array(np.float64, dp.n)))
    assert out[0][:2] == (5, 12)
    for o in out[1:]:
        assert o[:5] == out[0][:5] and np.array_equal(o[5], out[0][5])


def test_scale_properties_of_one_rank_of_config4_at_1000_per_row():
    """SURVEY 8d asks config 4 at r = 100 AND r = 1000 nonzeros per row.  At 1000 the whole matrix (1e10 nonzeros)
    only exists sharded: this is ONE rank's block of it at N = 8 -- rows [0, 1.25M) of the 10M x 10M system,
    1.25e9 nonzeros, 15 GB each for A_p and A_p' -- the largest matrix a rank of the scaling run can meet and the
    closest a real run comes to the 2^32 nonzeros a handle can index.  Generated in HBM; size-independent checks as
    for the other full-size shapes: acheck's adjoint identity, linearity, bit-level determinism of both products, a
    short solve that repeats itself exactly across the launch schedules and passes the reference's xcheck."""
    import torch
    from lsqr_amd.capi import DeviceBuffer
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~90 GB of free HBM while it builds")
    dp = devgen.generate("random:10000000:10000000:1000", 0, 1250000, itnlim=8)
    s = dp.solver
    info = s.info()
    assert dp.nnz == 1_250_000_000 and dp.nrows == 1_250_000 and dp.n == 10_000_000
    assert info["xlds"] == 3 and info["xlds_t"] == 3
    peak, kept = s.get_option("build_peak_bytes"), s.get_option("build_kept_bytes")
    print(f"build peak {peak / 1e9:.1f} GB beyond the 20 GB of triplets, kept {kept / 1e9:.1f} GB")
    assert kept < 33e9 and peak < 45e9
    inform, err = s.acheck()
    assert inform == 0 and err < 1e-12
    xa = P.u64_to_unit(P.rng_u64(1, 9, np.arange(dp.n, dtype=np.uint64)))
    yb = P.u64_to_unit(P.rng_u64(2, 9, np.arange(dp.nrows, dtype=np.uint64)))
    y1, y2, y3 = np.zeros(dp.nrows), np.zeros(dp.nrows), np.zeros(dp.nrows)
    s.aprod(1, dp.nrows, dp.n, xa, y1)
    s.aprod(1, dp.nrows, dp.n, xa, y2)
    s.aprod(1, dp.nrows, dp.n, 3.0 * xa, y3)
    assert np.array_equal(y1, y2)
    assert np.max(np.abs(y3 - 3.0 * y1)) <= 1e-13 * np.max(np.abs(y3))
    x1, x2 = np.zeros(dp.n), np.zeros(dp.n)
    s.aprod(2, dp.nrows, dp.n, x1, yb)
    s.aprod(2, dp.nrows, dp.n, x2, yb)
    assert np.array_equal(x1, x2)
    d_x = DeviceBuffer(8 * dp.n)
    out = []
    for pipeline in (2, 1, 0):
        s.set_option("pipeline", pipeline)
        r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
        out.append((r.istop, r.itn, r.anorm, r.rnorm, r.xnorm, d_x.to_array(np.float64, dp.n)))
    assert out[0][:2] == (5, 8)
    for o in out[1:]:
        assert o[:5] == out[0][:5] and np.array_equal(o[5], out[0][5])
    # 8 iterations do not converge; the reference's xcheck recomputes the residual u = b - A x from scratch (one more
    # mode-1 product of the 1.25e9 nonzeros): it must agree with the rnorm the recurrences carried along
    b = dp.d_b.to_array(np.float64, dp.nrows)
    inform, tests, u, v, w = s.xcheck(out[0][2], 1e-3, b, out[0][5])
    rn = np.sqrt(np.linalg.norm(u) ** 2 + (1e-3 * np.linalg.norm(out[0][5])) ** 2)
    assert np.isfinite(tests).all() and abs(rn - out[0][3]) <= 1e-8 * out[0][3]


def test_x_of_a_device_resident_solve_is_complete_when_solve_returns():
    """on a column-swept matrix (no fused schedule) k_out_copy follows the snapshot kernel, so a host
    that returns on the snapshot's seal returns before x has reached the caller's buffer.  The solve must only return once
    the copy is done: x read on ANOTHER stream (hipMemcpy on the null stream does not wait for the handle's non-blocking
    stream) right after lsqrhip_solve_device -- the stop in the last possible batch, spin poll at its default -- is the
    x a device-wide synchronise shows later.  n = 12M: the copy out is 96 MB each way."""
    from lsqr_amd import capi
    from lsqr_amd.capi import DeviceBuffer
    assert os.environ.get("LSQRHIP_SPIN_POLL", "1") != "0"
    old = os.environ.get("LSQRHIP_CSB")
    os.environ["LSQRHIP_CSB"] = "1"
    try:
        dp = devgen.generate("random:300000:12000000:40", itnlim=4)
    finally:
        os.environ.pop("LSQRHIP_CSB", None)
        if old is not None:
            os.environ["LSQRHIP_CSB"] = old
    s = dp.solver
    assert s.info()["xlds"] == 3                      # column-swept row blocks: the schedule is not the fused one
    s.atol = s.btol = s.conlim = 0.0
    d_x = DeviceBuffer(8 * dp.n)
    tail = 4096
    nan_tail = np.full(tail, np.nan)
    for k in range(6):
        # poison the END of x (what the copy kernel's last workgroups write), not by a full upload: keep the gap short
        capi.check(capi.lib().lsqrhip_dev_upload(d_x.ptr.value + 8 * (dp.n - tail), nan_tail.ctypes.data, 8 * tail))
        r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, 1e-3)
        got = np.empty(tail)
        capi.check(capi.lib().lsqrhip_dev_download(got.ctypes.data, d_x.ptr.value + 8 * (dp.n - tail), 8 * tail))
        capi.check(capi.lib().lsqrhip_dev_sync())
        later = np.empty(tail)
        capi.check(capi.lib().lsqrhip_dev_download(later.ctypes.data, d_x.ptr.value + 8 * (dp.n - tail), 8 * tail))
        assert (r.istop, r.itn) == (5, 4)
        assert not np.isnan(later).any()
        assert np.array_equal(got, later), k
