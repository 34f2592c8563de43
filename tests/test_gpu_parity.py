"""Parity of the HIP path (through the C-ABI) with the reference -- GPU only.

Bar (BASELINE.json north_star): x, anorm, rnorm within 1e-10 relative, istop identical.
Per case the tolerance is max(1e-10, 10 * band) where `band` is the reference's own
drift under a permutation of its COO input (recorded in tests/golden/solve_cases.json by
gen_golden.py): LSQR amplifies rounding over hundreds of iterations, so the reference
does not reproduce ITSELF to 1e-10 there.
"""
import json
import os

import numpy as np
import pytest

import oracle
from cases import TRUNCATED_TWINS, assert_log_lines_match, build_cases
from golden.gen_golden import blas_vectors
from lsqr_amd import capi, problems as P
from lsqr_amd.capi import LsqrHipError
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SOLVE = json.load(open(os.path.join(GOLD, "solve_cases.json")))
BLAS = json.load(open(os.path.join(GOLD, "blas1.json")))
CASES = build_cases()
TOL = 1e-10


def fh(s):
    return float.fromhex(s)


def fhv(lst):
    return np.array([float.fromhex(t) for t in lst], dtype=np.float64)


def make(p, o):
    return lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=o["atol"], btol=o["btol"],
                                       conlim=o["conlim"], itnlim=o["itnlim"])


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def test_extension_is_loaded_and_device_present():
    assert os.path.exists(capi.LIB_PATH)
    assert capi.device_count() >= 1


@pytest.mark.parametrize("name", sorted(CASES))
def test_solve_parity_vs_reference_golden(name):
    p, o = CASES[name]
    g = SOLVE[name]
    r = make(p, o).solve(p.b, o["damp"], wantse=o["wantse"])
    sens = g.get("sens", dict(x=0.0, anorm=0.0, rnorm=0.0, itn=[g["itn"]], istop=[g["istop"]]))
    assert r.istop == g["istop"], "istop must be identical"
    gx = fhv(g["x"])
    nx = np.linalg.norm(gx)
    tx = max(TOL, 10 * sens["x"])
    if nx == 0:
        assert np.all(r.x == 0.0)
    elif tx <= 1e-6:
        assert np.linalg.norm(r.x - gx) / nx <= tx
    else:
        # The reference's OWN x moves by more than 1e-7 when its COO input is permuted (illcond_conlim: 7e-3,
        # powerlaw_small: 4e-7, ...): a bound of ten times that asserts next to nothing, so none is pretended
        # here.  The strict 1e-10 on the same system lives in its truncated twin (cut where the reference's
        # drift is still < 1e-11: test_truncated_twins_hold_the_strict_tolerance); this run only has to stop
        # for the same reason, within the reference's spread of iterations, with a finite x of the right size.
        twins = [t for t in TRUNCATED_TWINS if t.startswith(name + "_it")]
        assert twins, f"{name}: a case this sensitive needs a truncated twin that carries the strict tolerance"
        assert np.all(np.isfinite(r.x)) and 0.1 * nx <= np.linalg.norm(r.x) <= 10 * nx
        # ... and wherever ten times the reference's own drift still says something about the direction of x
        # (below 1e-2: powerlaw_small 4e-6, ...), that bound is held as before; only the truly chaotic runs
        # (illcond_conlim: 7e-2) are left to the magnitude check and their twins.
        if tx <= 1e-2:
            assert np.linalg.norm(r.x - gx) / nx <= tx
    # itn is pinned only where the reference's own iteration count AND its own stopping
    # quantities do not move when its COO input is permuted (12 permutations, gen_golden.py):
    # anorm enters every stopping test (src/lsqr.f90:759-790), so a run whose anorm drifts by
    # 1e-4 under a permutation has no well-defined crossing iteration.
    pinned = len(sens["itn"]) == 1 and max(sens["anorm"], sens["rnorm"]) <= 1e-6
    if pinned:
        assert r.itn == g["itn"]
    else:
        # one more perturbation (ours) may land just outside the sampled spread: allow its width
        w = max(1, max(sens["itn"]) - min(sens["itn"]))
        assert min(sens["itn"]) - w <= r.itn <= max(sens["itn"]) + w
    stable = r.itn == g["itn"]              # same iteration -> the norms are comparable
    if stable:
        assert rel(r.anorm, fh(g["anorm"])) <= max(TOL, 10 * sens["anorm"]) or fh(g["anorm"]) == 0.0
        if g["rnorm"] is not None:
            grn = fh(g["rnorm"])
            # a residual at rounding level (rnorm/bnorm ~ eps) has no significant digits
            assert rel(r.rnorm, grn) <= max(TOL, 10 * sens["rnorm"]) or grn <= 1e-13 * np.linalg.norm(p.b)
        assert rel(r.xnorm, fh(g["xnorm"])) <= max(1e-9, 10 * sens["x"]) or fh(g["xnorm"]) == 0.0
    if g["rnorm"] is None:                   # istop = 0: rnorm defined as norm(b) (documented fix)
        assert r.rnorm == pytest.approx(np.linalg.norm(p.b), rel=1e-14)
    if o["wantse"]:
        gse = fhv(g["se"])
        if stable and fh(g["rnorm"]) > 1e-13 * np.linalg.norm(p.b):
            assert np.linalg.norm(r.se - gse) <= max(1e-9, 10 * sens["x"]) * np.linalg.norm(gse) + 1e-300
        else:
            # se = rnorm * sqrt(sigma/t) (src/lsqr.f90:857-865): with rnorm at rounding level
            # only the shape is meaningful -- compare se / rnorm
            assert np.allclose(r.se / r.rnorm, gse / fh(g["rnorm"]), rtol=1e-6)


@pytest.mark.parametrize("name", ["t1_readme_default", "t2_ez_3x4"])
def test_reference_ez_test_criterion(name):
    """test/lsqrtest_ez.f90:50,102: |A x - b| <= 1e-12 ; README.md:55-58."""
    p, o = CASES[name]
    r = make(p, o).solve(p.b, 0.0)
    assert r.istop == 1
    assert np.max(np.abs(p.dense() @ r.x - p.b)) <= 1e-12
    if name == "t1_readme_default":
        assert np.allclose(r.x, [1.242424, -6.060606e-2, -4.040404e-2], rtol=0, atol=5e-7)


@pytest.mark.parametrize("name", sorted(k for k, v in CASES.items() if v[0].nnz > 0))
def test_aprod_parity(name):
    """aprod mode 1 / 2 vs the reference's outputs; short rows are summed in COO order,
    so most entries agree bit for bit; all to a few ulp."""
    p, o = CASES[name]
    g = SOLVE[name]
    s = make(p, o)
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    assert np.array_equal(x, xp)
    g1 = fhv(g["aprod1_y"])
    assert np.max(np.abs(y - g1)) <= 1e-13 * max(np.max(np.abs(g1)), 1.0)
    x, y = xp.copy(), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    assert np.array_equal(y, yp)
    g2 = fhv(g["aprod2_x"])
    assert np.max(np.abs(x - g2)) <= 1e-13 * max(np.max(np.abs(g2)), 1.0)


@pytest.mark.parametrize("name", ["poisson_20x20_it50", "random_over_damped", "shuffled_dups", "itnlim_1"])
def test_aprod_bit_exact_on_short_rows(name):
    """Rows shorter than the 64-lane split threshold are reduced left to right in COO
    order: identical bits to the reference's row sums."""
    p, o = CASES[name]
    g = SOLVE[name]
    s = make(p, o)
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    assert np.array_equal(y, fhv(g["aprod1_y"]))


@pytest.mark.parametrize("name", sorted(k for k, v in CASES.items() if v[0].nnz > 0))
def test_acheck_xcheck_on_device_operator(name):
    p, o = CASES[name]
    g = SOLVE[name]
    s = make(p, o)
    inform, err = s.acheck()
    assert inform == g["acheck_inform"] == 0
    assert err <= 1e-13
    gx = fhv(g["x"])
    inform, tests, u, v, w = s.xcheck(fh(g["anorm"]), o["damp"], p.b, gx)
    assert inform == g["xcheck"]["inform"]
    gt = fhv(g["xcheck"]["tests"])
    # r = b - A x cancels: its relative accuracy is eps * (|b| + |A||x|) / |r| in BOTH
    # implementations, and test2/test3 inherit it
    rho1 = fh(g["xcheck"]["u_norm"])
    amp = (np.linalg.norm(p.b) + fh(g["anorm"]) * np.linalg.norm(gx)) / max(rho1, 1e-300)
    tol = 1e-9 + 1e-14 * amp
    # A'r of a least-squares solution cancels too: sigma1 is accurate to eps*|A||r|, i.e.
    # test2/test3 carry an ABSOLUTE error ~1e-14
    assert np.all(np.abs(tests - gt) <= tol * np.abs(gt) + 1e-14) or tol > 1e-2
    assert rel(np.linalg.norm(u), rho1) <= tol or tol > 1e-2


def test_error_behaviour_matches_reference_strings():
    p = P.readme_3x3()
    with pytest.raises(LsqrHipError) as e:
        lsqr_solver_ez().initialize(3, 3, p.a, np.array([1, 2, 4, 1, 2, 3, 1, 2, 3], np.int32), p.icol)
    assert e.value.code == 2 and e.value.message == "invalid irow or m in initialize_ez"
    with pytest.raises(LsqrHipError) as e:
        lsqr_solver_ez().initialize(3, 3, p.a, p.irow, np.array([1, 1, 1, 2, 2, 2, 3, 3, 7], np.int32))
    assert e.value.code == 3 and e.value.message == "invalid icol or n in initialize_ez"
    with pytest.raises(LsqrHipError) as e:
        lsqr_solver_ez().initialize(3, 3, p.a[:8], p.irow, p.icol)
    assert e.value.code == 1 and e.value.message == "invalid a,icol,irow sizes in initialize_ez"
    s = lsqr_solver_ez().initialize(3, 3, p.a, p.irow, p.icol)
    with pytest.raises(LsqrHipError) as e:
        s.aprod(3, 3, 3, np.zeros(3), np.zeros(3))
    assert e.value.code == 5 and e.value.message == "invalid mode in aprod_ez"
    with pytest.raises(LsqrHipError) as e:
        s.aprod(1, 4, 3, np.zeros(3), np.zeros(4))
    assert e.value.code == 4 and e.value.message == "lsqr_solver_ez class not properly initialized"
    with pytest.raises(LsqrHipError) as e:
        lsqr_solver_ez().solve(np.zeros(3), 0.0)
    assert e.value.code == 4


def test_reinitialize_resets_options_like_intent_out():
    """`me` is intent(out) in initialize_ez (src/lsqr.f90:95): defaults come back."""
    p = P.readme_3x3()
    s = lsqr_solver_ez().initialize(3, 3, p.a, p.irow, p.icol, atol=1e-3, itnlim=7)
    assert (s.atol, s.itnlim) == (1e-3, 7)
    s.initialize(3, 3, p.a, p.irow, p.icol)
    assert (s.atol, s.btol, s.conlim, s.itnlim, s.nout) == (0.0, 0.0, 0.0, 100, 0)
    assert s.solve(p.b, 0.0).istop == 1


def test_b_is_not_modified_and_solver_is_reusable():
    p, o = CASES["random_over_damped"]
    s = make(p, o)
    b = p.b.copy()
    r1 = s.solve(b, o["damp"])
    assert np.array_equal(b, p.b)                       # solve never modifies b (src/lsqr.f90:242)
    r2 = s.solve(b, o["damp"])
    assert np.array_equal(r1.x, r2.x) and r1.anorm == r2.anorm and r1.itn == r2.itn   # run-to-run determinism
    r3 = s.solve(2.0 * b, o["damp"])
    assert np.allclose(r3.x, 2.0 * r1.x, rtol=1e-12, atol=0)


def test_graph_and_eager_launch_modes_agree_bitwise():
    p, o = CASES["poisson_20x20_it50"]
    s = make(p, o)
    r_graph = s.solve(p.b, 0.0)
    s.set_option("graph", 0)
    r_eager = s.solve(p.b, 0.0)
    s.set_option("time_kernels", 1)
    r_timed = s.solve(p.b, 0.0)
    t = s.last_timing()
    for r in (r_eager, r_timed):
        assert np.array_equal(r.x, r_graph.x) and r.itn == r_graph.itn and r.anorm == r_graph.anorm
    assert t.spmv1_launches == r_graph.itn and t.spmv1_ms > 0.0
    s.set_option("time_kernels", 0)
    s.set_option("graph", 1)
    s.set_option("graph_iters", 7)                     # batch size must not change anything
    r7 = s.solve(p.b, 0.0)
    assert np.array_equal(r7.x, r_graph.x) and r7.itn == r_graph.itn


@pytest.mark.parametrize("name", ["poisson_20x20_it50", "random_over_se", "illcond_conlim", "itnlim_1", "b_zero"])
def test_look_ahead_poll_changes_nothing(name):
    """solve_loop.h: with graphs the host enqueues batch k+1 before it waits for batch k's stop
    flag.  A stop in the middle of a batch, at a batch boundary, before the first batch ends, or
    an iteration limit that is not a multiple of the batch must give the strict loop's answer."""
    p, o = CASES[name]
    s = make(p, o)
    s.set_option("poll_ahead", 0)
    ref = s.solve(p.b, o["damp"], wantse=o["wantse"])
    for gi in (2, 4, 6, 16, max(2, ref.itn), max(2, ref.itn + 1), 200):
        for pa in (1, 0):
            s.set_option("graph_iters", gi)
            s.set_option("poll_ahead", pa)
            r = s.solve(p.b, o["damp"], wantse=o["wantse"])
            assert (r.istop, r.itn, r.anorm, r.acond, r.rnorm, r.arnorm, r.xnorm) == \
                   (ref.istop, ref.itn, ref.anorm, ref.acond, ref.rnorm, ref.arnorm, ref.xnorm), (gi, pa)
            assert np.array_equal(r.x, ref.x)
            if o["wantse"]:
                assert np.array_equal(r.se, ref.se)
    # the limit reached exactly at / one past / one short of a batch boundary
    for itnlim in (7, 8, 9):
        out = []
        for pa in (0, 1):
            s.set_option("graph_iters", 8)
            s.set_option("poll_ahead", pa)
            s.itnlim = itnlim
            s.atol = s.btol = s.conlim = 0.0
            r = s.solve(p.b, o["damp"])
            out.append((r.istop, r.itn, r.anorm, r.rnorm, r.x.copy()))
        assert out[0][:4] == out[1][:4] and np.array_equal(out[0][4], out[1][4])


@pytest.mark.parametrize("name", ["poisson_20x20_it50", "random_over_se", "powerlaw_small", "illcond_conlim",
                                  "empty_rows_cols", "t1_readme_damped", "b_zero"])
def test_pipelined_and_sequential_schedules_agree_bitwise(name):
    """solve_loop.h: the rider schedule (SpMVs derive their own norm, scalar machine inside the
    next SpMV launch) and the fused-update schedule (x/w update inside the mode-1 launch, every
    workgroup recomputing the rotation) must reproduce the plain sequential schedule bit for
    bit -- same x, se, scalars, istop, itn -- in graph and in eager mode."""
    p, o = CASES[name]
    s = make(p, o)
    ref = None
    for pipeline in (0, 1, 2):
        for graph in (1, 0):
            s.set_option("pipeline", pipeline)
            s.set_option("graph", graph)
            r = s.solve(p.b, o["damp"], wantse=o["wantse"])
            key = (r.istop, r.itn, r.anorm, r.acond, r.rnorm, r.arnorm, r.xnorm)
            if ref is None:
                ref = (r, key)
            else:
                assert key == ref[1], (pipeline, graph)
                assert np.array_equal(r.x, ref[0].x)
                if o["wantse"]:
                    assert np.array_equal(r.se, ref[0].se)


@pytest.mark.parametrize("name", sorted(BLAS))
def test_device_blas1_vs_reference_golden(name):
    x = blas_vectors()[name]
    g = BLAS[name]
    p = P.readme_3x3()
    s = lsqr_solver_ez().initialize(3, 3, p.a, p.irow, p.icol)
    L = capi.lib()
    import ctypes as C
    n = len(x)
    dx = capi.DeviceBuffer.from_array(x if n else np.zeros(1))
    out = C.c_double()
    capi.check(L.lsqrhip_dnrm2(s._h, n, dx.ptr, C.addressof(out)))
    want = fh(g["dnrm2"])
    # the stand-alone device dnrm2 is scaled like the reference's (huge / tiny / mixed vectors too)
    assert rel(out.value, want) <= 1e-14 or want == 0.0 == out.value
    if n:
        y = x[::-1].copy()
        dy = capi.DeviceBuffer.from_array(y)
        capi.check(L.lsqrhip_ddot(s._h, n, dx.ptr, dy.ptr, C.addressof(out)))
        wd = fh(g["ddot_rev"])
        if np.isfinite(wd):
            assert abs(out.value - wd) <= 1e-13 * max(np.sum(np.abs(x * y)), 1e-300)
        if "dscal_m037" in g:
            capi.check(L.lsqrhip_dscal(s._h, n, -0.37, dx.ptr))
            assert np.array_equal(dx.to_array(np.float64, n), fhv(g["dscal_m037"]))     # bit exact
        capi.check(L.lsqrhip_dcopy(s._h, n, dy.ptr, dx.ptr))
        assert np.array_equal(dx.to_array(np.float64, n), y)


@pytest.mark.parametrize("name", ["t1_readme_default", "random_over_damped"])
def test_iteration_log_text_matches_reference(name, tmp_path):
    """nout /= 0: the log formatted from device records equals the reference's log
    (tests/golden/log_*.txt) except where a printed value sits on a rounding boundary."""
    p, o = CASES[name]
    path = str(tmp_path / "log.txt")
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=o["atol"], btol=o["btol"],
                                    conlim=o["conlim"], itnlim=o["itnlim"], nout=path)
    s.solve(p.b, o["damp"])
    got = open(path).read().splitlines()
    want = open(os.path.join(GOLD, f"log_{name}.txt")).read().splitlines()
    assert len(got) == len(want)
    same = sum(a == b for a, b in zip(got, want))
    if name == "random_over_damped":
        assert same >= len(want) - 2
    # structure (headers, exit block, which iterations are printed) is identical
    assert [l[:6] for l in got] == [l[:6] for l in want]


@pytest.mark.parametrize("name", TRUNCATED_TWINS)
def test_truncated_twins_hold_the_strict_tolerance(name, tmp_path):
    """The four long runs whose tolerance above is widened by the reference's own drift (illcond_conlim,
    powerlaw_small, poisson_48x37_tol, empty_rows_cols), stopped where that drift is still < 1e-11: strict
    1e-10 on x, anorm, rnorm with identical istop and itn, through all three launch schedules -- and every
    printed line of the iteration log (x(1), rnorm to their 10 digits; test1, test2, anorm, acond to their
    3; phi, dknorm, dxk, alfa_opt to their 2) against the reference's own log of the same run."""
    import re
    p, o = CASES[name]
    g = SOLVE[name]
    sens = g["sens"]
    assert max(sens["x"], sens["anorm"], sens["rnorm"]) < 1e-11 and sens["itn"] == [g["itn"]]
    gx = fhv(g["x"])
    path = str(tmp_path / "log.txt")
    for pipeline in (2, 1, 0):
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=o["atol"], btol=o["btol"],
                                        conlim=o["conlim"], itnlim=o["itnlim"], nout=path if pipeline == 2 else None)
        s.set_option("pipeline", pipeline)
        r = s.solve(p.b, o["damp"])
        assert (r.istop, r.itn) == (g["istop"], g["itn"])
        assert np.linalg.norm(r.x - gx) <= TOL * np.linalg.norm(gx)
        assert rel(r.anorm, fh(g["anorm"])) <= TOL and rel(r.rnorm, fh(g["rnorm"])) <= TOL
        assert rel(r.xnorm, fh(g["xnorm"])) <= TOL and rel(r.acond, fh(g["acond"])) <= 1e-9
    got = open(path).read().splitlines()
    want = open(os.path.join(GOLD, f"log_{name}.txt")).read().splitlines()
    assert_log_lines_match(got, want, min(g["itn"], 10))


def test_log_holds_every_line_of_a_run_that_lingers_near_convergence(tmp_path):
    """The reference prints EVERY iteration whose tests are within a factor 10 of their tolerances
    (src/lsqr.f90:815-822): 272 lines for the 402 iterations of this run.  The device keeps its records
    in a buffer sized from itnlim: none may be dropped, and the stopping iteration is the last line."""
    p = P.poisson2d(48, 37)
    path = str(tmp_path / "log.txt")
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=1e-4, btol=1e-4, itnlim=1300, nout=path)
    r = s.solve(p.b, 0.0)
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, atol=1e-4, btol=1e-4, itnlim=1300)
    assert (r.istop, r.itn) == (o.istop, o.itn) == (1, 402)
    rec = s.log_records()
    assert s.get_option("log_truncated") == 0
    # (the reference prints 272; the iteration at which test1 first drops below 10 rtol is decided at
    # rounding level, so one line more or less at the entry of the band is legitimate)
    assert 270 <= len(rec) <= 274 and int(rec[-1][0]) == r.itn and int(rec[-1][11]) == 1
    assert list(rec[:10, 0]) == list(range(1, 11))
    assert np.all(np.diff(rec[:, 0]) > 0)                       # ascending iterations, none twice
    lines = [l for l in open(path).read().splitlines() if l[:6].strip().isdigit()]
    assert len(lines) == len(rec) + 1                            # + the line of iteration 0


def test_full_size_config2_poisson_vs_oracle():
    """BASELINE.json configs[1]: 1M x 1M 5-point Poisson, damp = 0.  200 iterations against the
    CPU checker (~3 s), plus the size-independent checks acheck / linearity."""
    p = P.poisson2d(1000, 1000)
    assert p.nnz == 4_996_000
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=200)
    r = s.solve(p.b, 0.0)
    o = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, itnlim=200)
    assert (r.istop, r.itn) == (o.istop, o.itn) == (5, 200)
    assert np.linalg.norm(r.x - o.x) <= TOL * np.linalg.norm(o.x)
    assert rel(r.anorm, o.anorm) <= TOL and rel(r.rnorm, o.rnorm) <= TOL
    inform, err = s.acheck()
    assert inform == 0 and err < 1e-12
    xa = P.u64_to_unit(P.rng_u64(1, 9, np.arange(p.n, dtype=np.uint64)))
    xb = P.u64_to_unit(P.rng_u64(2, 9, np.arange(p.n, dtype=np.uint64)))
    ya, yb, yab = np.zeros(p.m), np.zeros(p.m), np.zeros(p.m)
    s.aprod(1, p.m, p.n, xa, ya)
    s.aprod(1, p.m, p.n, xb, yb)
    s.aprod(1, p.m, p.n, xa + 2.0 * xb, yab)
    assert np.max(np.abs(yab - (ya + 2.0 * yb))) <= 1e-13 * np.max(np.abs(yab))


def test_powerlaw_long_rows_exercise_all_three_phases():
    """config 5 shape at test scale: rows from 4 to 6000 nonzeros (long-row split path).
    The dense rows give A one huge singular value and the Golub-Kahan vectors lose
    orthogonality fast: the REFERENCE run on a permuted copy of its own input drifts
    1e-15 / 4e-12 / 1e-9 / 2.5e-5 in x at 6 / 15 / 20 / 30 iterations (measured with
    oracle/_ref).  So the 1e-10 bar is checked at 12 iterations."""
    p = P.powerlaw_rows(20000, 8000, seed=21, dmin=4, dmax=6000)
    assert np.max(np.bincount(p.irow)) >= 6000
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=12)
    po = oracle.port()
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    yp = P.u64_to_unit(P.rng_u64(102, 9, np.arange(p.m, dtype=np.uint64)))
    x, y = xp.copy(), yp.copy()
    s.aprod(1, p.m, p.n, x, y)
    _, y_ref = po.aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(y - y_ref)) <= 1e-12 * np.max(np.abs(y_ref))
    x, y = xp.copy(), yp.copy()
    s.aprod(2, p.m, p.n, x, y)
    x_ref, _ = po.aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
    assert np.max(np.abs(x - x_ref)) <= 1e-12 * np.max(np.abs(x_ref))
    r = s.solve(p.b, 0.0)
    o = po.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, itnlim=12)
    assert (r.istop, r.itn) == (o.istop, o.itn)
    assert np.linalg.norm(r.x - o.x) <= TOL * np.linalg.norm(o.x)
    assert rel(r.anorm, o.anorm) <= TOL and rel(r.rnorm, o.rnorm) <= TOL


def test_unsorted_and_sorted_coo_give_same_matrix():
    """K0: the radix-sort path (shuffled COO) and the already-sorted fast path build the same
    operator; only the within-row order (= summation order) differs."""
    p = P.random_rows(5000, 1500, 9, seed=17)
    q = P.shuffled(p, 5)
    sp = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=20)
    sq = lsqr_solver_ez().initialize(q.m, q.n, q.a, q.irow, q.icol, itnlim=20)
    xp = P.u64_to_unit(P.rng_u64(101, 9, np.arange(p.n, dtype=np.uint64)))
    y1, y2 = np.zeros(p.m), np.zeros(p.m)
    sp.aprod(1, p.m, p.n, xp, y1)
    sq.aprod(1, q.m, q.n, xp, y2)
    assert np.max(np.abs(y1 - y2)) <= 1e-14 * np.max(np.abs(y1))
    # shuffled input: bit-exact against the reference run on the same shuffled triplets
    _, yr = oracle.port().aprod(1, q.m, q.n, q.irow, q.icol, q.a, xp, np.zeros(q.m))
    assert np.array_equal(y2, yr)


@pytest.mark.parametrize("name", ["poisson_20x20_it50", "random_over_se", "illcond_conlim", "itnlim_1", "b_zero",
                                  "zero_matrix", "one_by_one"])
def test_device_resident_solve_copies_x_out_in_graph(name):
    """lsqrhip_solve_device: x is copied to the caller's device array by the batch that raises the stop flag
    (vec.h k_out_copy), b is read where it lies through the pinned slot (k_start).  Whatever the batch size
    and wherever the stop falls -- mid-batch with a look-ahead batch behind it, at a boundary, at iteration 0 --
    the outputs must be those of the host-vector solve, bit for bit, and the caller's b must be untouched."""
    from lsqr_amd import capi
    p, o = CASES[name]
    s = make(p, o)
    ref = s.solve(p.b, o["damp"], wantse=o["wantse"])
    d_b = capi.DeviceBuffer.from_array(np.ascontiguousarray(p.b, np.float64))
    d_x = capi.DeviceBuffer(8 * max(p.n, 1))
    d_se = capi.DeviceBuffer(8 * max(p.n, 1))
    poison = np.full(max(p.n, 1), np.nan)
    for gi in (2, 6, max(2, ref.itn), max(2, ref.itn + 1), 64):
        for pa in (1, 0):
            for use_graph in (1, 0):
                s.set_option("graph", use_graph)
                s.set_option("graph_iters", gi)
                s.set_option("poll_ahead", pa)
                d_x.copy_from(poison)                                  # stale contents must be overwritten
                r = s.solve_device(d_b.ptr.value, d_x.ptr.value, o["damp"])
                assert (r.istop, r.itn, r.anorm, r.rnorm, r.xnorm) == \
                       (ref.istop, ref.itn, ref.anorm, ref.rnorm, ref.xnorm), (gi, pa, use_graph)
                assert np.array_equal(d_x.to_array(np.float64, p.n), ref.x), (gi, pa, use_graph)
                if o["wantse"]:
                    r = s.solve_device(d_b.ptr.value, d_x.ptr.value, o["damp"], d_se.ptr.value)
                    assert np.array_equal(d_x.to_array(np.float64, p.n), ref.x)
                    assert np.array_equal(d_se.to_array(np.float64, p.n), ref.se)
    # an output address that is aligned for a double but not for a pair of them (the fused batch tail stores x there
    # itself, element by element: spmv.h k_update_lazy copy_out)
    d_x2 = capi.DeviceBuffer(8 * (p.n + 1))
    d_x2.copy_from(np.full(p.n + 1, np.nan))
    s.set_option("graph", 1)
    s.set_option("graph_iters", 6)
    s.set_option("poll_ahead", 1)
    r = s.solve_device(d_b.ptr.value, d_x2.ptr.value + 8, o["damp"])
    got = d_x2.to_array(np.float64, p.n + 1)
    assert (r.istop, r.itn) == (ref.istop, ref.itn) and np.isnan(got[0]) and np.array_equal(got[1:], ref.x)
    assert np.array_equal(d_b.to_array(np.float64, p.m), p.b)
