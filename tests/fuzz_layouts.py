#!/usr/bin/env python3
"""Random small systems through every layout (forced by the knobs) against the oracle's aprod and a
short solve.  Edge cases on purpose: empty rows / columns, one very long row, m < n, m > n, nnz = 0,
duplicates, dictionary and non-dictionary values.  The generator and the acceptance rule of
tests/test_gpu_fuzz.py; `python tests/fuzz_layouts.py [ncases] [seed] [--bands] [--engine | --real32]` runs more cases (--engine: the
same cases through the sharded engine, several ranks on the one device; --real32: every layout as a REAL32 handle)
(--bands prints every case's measured bands: profiles/r04/fuzz_bands.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from lsqr_amd.solver import lsqr_solver_ez

# How far a short solve may lie from the oracle's where a row (or a column, mode 2) is longer than 16 -- summed by
# several lanes, a tree instead of the reference's left-to-right sum -- and the norms are tree sums too.  Measured
# per case on the REFERENCE itself (the oracle), nothing is exempt:
#   (a) its x under six permutations of its COO input (the order of a row sum is the only freedom it leaves there);
#   (b) its x with ONE norm of its bidiagonalisation moved by ONE unit in the last place (oracle_set_norm_ulp: every
#       beta and alpha of the run in turn, both directions) -- the rounding a permutation does not touch;
#   (c) its x with ACCURATE sums (compensated row sums, pairwise norms: oracle.set_accurate_sums) -- the same
#       recurrences, correctly rounded.  This is where the reference's own rounding error shows in full: a row of
#       thousands of EQUAL addends (dictionary values) drifts the same way in every order -- 64 x 5000 with one
#       6000-entry row: 2.8e-14 of the row sum in all permutations, 130 ulps of anorm in the FIRST iteration, and,
#       amplified ~500 x per iteration by that system, 23 % of x after six (profiles/r04/fuzz_bands.txt).  The GPU
#       sums in trees and exact integers: it follows (c), digit for digit where the reference does not.
# The GPU's x must lie within BAND_FACTOR x max((a), (b)) -- at least TIGHT -- of the reference's x OR of (c): of the
# reference evaluated in one of two legal summation orders.  The share of results that needed more than TIGHT is
# counted and bounded: it is a property of the generator (one case in six has a 6000-entry row, about half of those
# amplify rounding this much in six iterations), measured at 7-8 % over seeds 1-3, 73, 81.
# The standard errors se of the same solve are held to the same tolerance; where they need more -- a system exhausted
# before the sixth step adds the (w / rho)^2 of iterations that run on rounding noise, which x does not feel -- the band
# is measured on the reference's OWN se (permutations, one ulp of one norm) and that share is counted and bounded too.
BAND_FACTOR = 200.0
N_PERMUTATIONS = 6
TIGHT = 1e-9
MAX_WIDENED_SHARE = 0.10

KNOBS = ["LSQRHIP_SELL", "LSQRHIP_SELLP", "LSQRHIP_VAL8", "LSQRHIP_COL16", "LSQRHIP_PANELS", "LSQRHIP_PANEL_KB",
         "LSQRHIP_XLDS", "LSQRHIP_XLDS_COLS", "LSQRHIP_OFF64", "LSQRHIP_SKEW", "LSQRHIP_TUNE", "LSQRHIP_CSB", "LSQRHIP_CSB_R", "LSQRHIP_CSB_S", "LSQRHIP_CSB_NARROW", "LSQRHIP_PAT", "LSQRHIP_PAT2", "LSQRHIP_PAT_PAIR", "LSQRHIP_SPAT"]
LAYOUTS = [
    {},                                                                        # whatever the build chooses
    {"LSQRHIP_PAT": "1"},                                                      # row patterns whenever the limits hold (pat.h)
    {"LSQRHIP_PAT": "1", "LSQRHIP_PAT_PAIR": "0"},                             # ... in the slice form (lane L owns row L) instead of paired rows
    {"LSQRHIP_PAT": "1", "LSQRHIP_PAT2": "1"},                                 # ... up to 4096 of them (two-byte pattern numbers)
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "1"},                                 # structure patterns whenever the limits hold
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0"},                                 # ... and neither kind
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_SELL": "0"},
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_PANELS": "1", "LSQRHIP_PANEL_KB": "64", "LSQRHIP_XLDS": "0"},    # L2 panels (8192 columns)
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_PANELS": "1", "LSQRHIP_PANEL_KB": "64", "LSQRHIP_XLDS": "0", "LSQRHIP_OFF64": "1", "LSQRHIP_SKEW": "0"},
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_XLDS": "1", "LSQRHIP_XLDS_COLS": "1024"},                        # LDS panels (1024 columns)
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_XLDS": "1", "LSQRHIP_XLDS_COLS": "1024", "LSQRHIP_COL16": "0", "LSQRHIP_OFF64": "1"},
    {"LSQRHIP_PAT": "0", "LSQRHIP_SPAT": "0", "LSQRHIP_SELL": "1", "LSQRHIP_SELLP": "0"},
    {"LSQRHIP_CSB": "1"},                                                      # column-swept row blocks (csb.h)
    {"LSQRHIP_CSB": "1", "LSQRHIP_CSB_R": "37"},                               # ... in many small blocks, ragged last one
    {"LSQRHIP_CSB": "1", "LSQRHIP_CSB_R": "129", "LSQRHIP_CSB_S": "3"},        # ... three workgroups per block (column splits)
    {"LSQRHIP_CSB": "1", "LSQRHIP_CSB_R": "700", "LSQRHIP_CSB_S": "2", "LSQRHIP_CSB_NARROW": "1"},   # ... 11-byte nonzeros where the deltas fit
]


# ... and the same cases through the sharded C++ engine (csrc/shard_engine.h), several ranks on the one device
# (LSQRHIP_SHARD_LOOPBACK=1: the exchanges as device copies), row blocks cut wherever the nonzeros put them: more ranks
# than rows, empty blocks, ragged column slices, the overlapped schedule.  (world, environment)
ENGINE_KNOBS = ["LSQRHIP_SHARD_LOOPBACK", "LSQRHIP_SHARD_OVERLAP", "LSQRHIP_SHARD_PARTS"]
ENGINES = [
    (2, {}),
    (3, {"LSQRHIP_CSB": "1", "LSQRHIP_CSB_R": "129"}),
    (5, {"LSQRHIP_SHARD_OVERLAP": "1", "LSQRHIP_CSB": "1"}),
    (8, {"LSQRHIP_SHARD_OVERLAP": "1", "LSQRHIP_SHARD_PARTS": "3"}),
]


class _ShardedHandle:
    """lsqrhip_create_sharded + the ordinary entry points on it, with the solver's call shapes"""

    def __init__(self, m, n, a, irow, icol, world):
        import ctypes as C
        from lsqr_amd.capi import check, lib
        self.C, self.check, self.lib, self.m, self.n = C, check, lib, m, n
        self.h = C.c_void_p()
        check(lib().lsqrhip_create_sharded(m, n, a.size, irow.ctypes.data, icol.ctypes.data, a.ctypes.data, world,
                                           C.byref(self.h)))

    def aprod(self, mode, x, y):
        self.check(self.lib().lsqrhip_aprod(self.h, mode, x.ctypes.data, y.ctypes.data))

    def solve(self, b, damp, itnlim):
        C = self.C
        x = np.zeros(max(self.n, 1))
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        self.check(self.lib().lsqrhip_solve(self.h, b.ctypes.data, damp, 0.0, 0.0, 0.0, itnlim, 0, 0, x.ctypes.data,
                                            None, C.addressof(istop), C.addressof(itn), *[C.addressof(s) for s in sc]))
        return x[:self.n], istop.value, itn.value

    def close(self):
        self.check(self.lib().lsqrhip_destroy(self.h))


def make_case(rs):
    kind = rs.randint(0, 6)
    m = int(rs.choice([1, 2, 63, 64, 65, 300, 1000, 5000, 20000]))
    n = int(rs.choice([1, 2, 63, 64, 65, 300, 1000, 5000, 30000]))
    if kind == 0:      # uniform sparse
        per = rs.randint(0, 12)
        irow = np.repeat(np.arange(m), per); icol = rs.randint(0, n, size=irow.size)
    elif kind == 1:    # banded (SELL territory)
        offs = rs.choice(np.arange(-5, 6), size=rs.randint(1, 6), replace=False)
        r = np.arange(m); rows = []; cols = []
        for o in offs:
            c = r * n // max(m, 1) + o
            ok = (c >= 0) & (c < n) & (rs.rand(m) > 0.05)
            rows.append(r[ok]); cols.append(c[ok])
        irow = np.concatenate(rows); icol = np.concatenate(cols)
    elif kind == 2:    # power law with a few very long rows
        deg = np.minimum((rs.pareto(1.2, size=m) * 3).astype(int), 4 * n)
        deg[rs.randint(0, m)] = min(6000, 4 * n)
        irow = np.repeat(np.arange(m), deg); icol = rs.randint(0, n, size=irow.size)
    elif kind == 3:    # empty matrix / nearly empty
        k = rs.randint(0, 3)
        irow = rs.randint(0, m, size=k); icol = rs.randint(0, n, size=k)
    elif kind == 4:    # dense-ish rows (LDS panel territory)
        per = rs.randint(50, 400)
        mm = min(m, 400)
        irow = np.repeat(np.arange(mm), per); icol = rs.randint(0, n, size=irow.size)
    else:              # a few dense columns + random
        k = rs.randint(1, 4000)
        irow = rs.randint(0, m, size=k); icol = np.where(rs.rand(k) < 0.3, rs.randint(0, min(n, 3), size=k), rs.randint(0, n, size=k))
    if rs.rand() < 0.5:
        a = rs.choice([-1.0, 4.0, 0.5, -0.0, 2.25], size=irow.size)            # dictionary
    else:
        a = rs.uniform(-1, 1, size=irow.size)
    perm = rs.permutation(irow.size) if rs.rand() < 0.5 else np.arange(irow.size)
    irow, icol, a = irow[perm], icol[perm], a[perm]
    b = rs.uniform(-1, 1, size=m)
    return m, n, (irow + 1).astype(np.int32), (icol + 1).astype(np.int32), a.astype(np.float64), b


def run(ncases, seed, verbose=True, bands=False, only=None, engine=False, real32=False):
    """Returns (failures, results that needed a tolerance above TIGHT, results in all).  engine: the sharded engine's
    variants (ENGINES) in place of the layouts.  real32: every layout as a REAL32 handle (src/lsqr_kinds.F90:16-17) --
    see run_real32."""
    if real32:
        return run_real32(ncases, seed, verbose)
    rs = np.random.RandomState(seed)
    po = oracle.port()
    bad = widened = total = se_wide = wide_hits = se_noise = 0
    for case in range(ncases):
        m, n, irow, icol, a, b = make_case(rs)
        xp, yp = rs.uniform(-1, 1, size=n), rs.uniform(-1, 1, size=m)
        _, y_ref = po.aprod(1, m, n, irow, icol, a, xp, yp)
        x_ref, _ = po.aprod(2, m, n, irow, icol, a, xp, yp)
        if only is not None and case not in only:
            continue
        o = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=6, wantse=True)
        longest = max(int(np.bincount(irow - 1, minlength=m).max()), int(np.bincount(icol - 1, minlength=n).max())) if irow.size else 0
        tol_long, band_perm, band_ulp, band_acc, o_acc = TIGHT, 0.0, 0.0, 0.0, None
        if longest > 16 and o.itn > 0:
            nx = max(np.linalg.norm(o.x), 1e-300)
            for k in range(1, 1 + N_PERMUTATIONS):
                perm = np.random.RandomState(1000 + k).permutation(irow.size)
                o2 = po.solve(m, n, irow[perm], icol[perm], a[perm], b, damp=1e-2, itnlim=6)
                band_perm = max(band_perm, float(np.linalg.norm(o2.x - o.x) / nx))
            try:
                for it in range(0, o.itn + 1):
                    for which in (1, 2):
                        for ulps in (1, -1):
                            po.set_norm_ulp(it, which, ulps)
                            o2 = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=6)
                            band_ulp = max(band_ulp, float(np.linalg.norm(o2.x - o.x) / nx))
            finally:
                po.set_norm_ulp(0, 0, 0)
            try:
                po.set_accurate_sums(True)
                o_acc = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=6, wantse=True)
            finally:
                po.set_accurate_sums(False)
            band_acc = float(np.linalg.norm(o_acc.x - o.x) / nx)
            tol_long = max(TIGHT, BAND_FACTOR * max(band_perm, band_ulp))
        if bands:
            print(f"case {case} m={m} n={n} nnz={irow.size} longest={longest} itn={o.itn} reference x under permutation "
                  f"{band_perm:.2e}, under one ulp of one norm {band_ulp:.2e}, with accurate sums {band_acc:.2e} "
                  f"-> tolerance {tol_long:.2e}", flush=True)
        worst = worst_ref = 0.0
        se_band = None
        o_prev = None

        def last_iteration_ran_on_noise():
            """The reference's own normal-equations residual one iteration BEFORE its last was already at rounding level
            (test2 = arnorm / (anorm rnorm) <= 1e-12, src/lsqr.f90:751-778): the Krylov space was exhausted, the last
            iteration's v is normalised noise.  The rule tests/test_gpu_parity.py:101-109 applies to its goldens, here from
            the oracle run one iteration shorter."""
            nonlocal o_prev
            if o.itn < 2:
                return False
            if o_prev is None:
                o_prev = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=o.itn - 1, wantse=True)
            den = o_prev.anorm * o_prev.rnorm
            return o_prev.itn == o.itn - 1 and (den == 0.0 or o_prev.arnorm <= 1e-12 * den)

        for lay in (ENGINES if engine else LAYOUTS):
            se_widened, se_ok, e4 = False, True, 0.0
            for k in KNOBS + ENGINE_KNOBS:
                os.environ.pop(k, None)
            sh = None
            try:
                if engine:
                    world, env = lay
                    os.environ.update(env)
                    os.environ["LSQRHIP_SHARD_LOOPBACK"] = "1"
                    sh = _ShardedHandle(m, n, a, irow, icol, world)
                    x, y = xp.copy(), yp.copy()
                    sh.aprod(1, x, y)
                    e1 = np.max(np.abs(y - y_ref)) / max(np.max(np.abs(y_ref)), 1.0)
                    x, y = xp.copy(), yp.copy()
                    sh.aprod(2, x, y)
                    e2 = np.max(np.abs(x - x_ref)) / max(np.max(np.abs(x_ref)), 1.0)
                    from types import SimpleNamespace
                    xs, istop_s, itn_s = sh.solve(b, 1e-2, 6)
                    r = SimpleNamespace(x=xs, istop=istop_s, itn=itn_s)
                    s = SimpleNamespace(info=lambda: {"world": world, **env})
                else:
                    os.environ.update(lay)
                    s = lsqr_solver_ez().initialize(m, n, a, irow, icol, itnlim=6)
                    x, y = xp.copy(), yp.copy()
                    s.aprod(1, m, n, x, y)
                    e1 = np.max(np.abs(y - y_ref)) / max(np.max(np.abs(y_ref)), 1.0)
                    x, y = xp.copy(), yp.copy()
                    s.aprod(2, m, n, x, y)
                    e2 = np.max(np.abs(x - x_ref)) / max(np.max(np.abs(x_ref)), 1.0)
                    r = s.solve(b, 1e-2, wantse=True)
                e3 = np.linalg.norm(r.x - o.x) / max(np.linalg.norm(o.x), 1e-300) if o.itn > 0 else float(np.max(np.abs(r.x)))
                e3_ref = e3
                if o_acc is not None:   # ... or the reference with accurate sums, whichever is nearer (header (c))
                    e3 = min(e3, float(np.linalg.norm(r.x - o_acc.x) / max(np.linalg.norm(o.x), 1e-300)))
                # The standard errors (src/lsqr.f90:857-865) ride on the same recurrences: same rule.  Their own band is
                # measured when they need one (lazily, once per case): a system exhausted after k < 6 steps runs its
                # last iteration on rounding noise, which x does not feel and se -- it adds (w / rho)^2 of that iteration
                # -- does; the reference's se under one ulp of one norm / a permutation says by how much.
                if not engine and r.se is not None and o.itn > 0 and r.itn == o.itn:
                    nse = max(float(np.linalg.norm(o.se)), 1e-300)
                    e4 = float(np.linalg.norm(r.se - o.se) / nse)
                    if o_acc is not None and o_acc.itn == r.itn:
                        e4 = min(e4, float(np.linalg.norm(r.se - o_acc.se) / nse))
                    if e4 >= 1e-12 and e4 >= tol_long:
                        if se_band is None:
                            se_band = 0.0
                            for k in range(1, 1 + N_PERMUTATIONS):
                                perm = np.random.RandomState(1000 + k).permutation(irow.size)
                                o2 = po.solve(m, n, irow[perm], icol[perm], a[perm], b, damp=1e-2, itnlim=6, wantse=True)
                                if o2.itn == o.itn:
                                    se_band = max(se_band, float(np.linalg.norm(o2.se - o.se) / nse))
                            try:
                                for it in range(0, o.itn + 1):
                                    for which in (1, 2):
                                        for ulps in (1, -1):
                                            po.set_norm_ulp(it, which, ulps)
                                            o2 = po.solve(m, n, irow, icol, a, b, damp=1e-2, itnlim=6, wantse=True)
                                            if o2.itn == o.itn:
                                                se_band = max(se_band, float(np.linalg.norm(o2.se - o.se) / nse))
                            finally:
                                po.set_norm_ulp(0, 0, 0)
                        if e4 < max(TIGHT, BAND_FACTOR * se_band):
                            se_widened = True      # accepted on the measured band: counted (se_wide) and bounded like x's
                        elif last_iteration_ran_on_noise():
                            # (round 6, seed 605: m = 2 / 5000, n = 65, TWO nonzeros -- solved exactly after two steps; the
                            #  third, which both sides run, normalises rounding noise into a unit vector and se adds
                            #  (w / rho)^2 of it, src/lsqr.f90:733-737.  The bands above are zero for two nonzeros.)  What is
                            #  left to hold: every se_j is at least what the iterations BEFORE the noise had accumulated --
                            #  the sums only grow -- and finite.
                            lb = o_prev.se * (o.rnorm / o_prev.rnorm) if o_prev.rnorm > 0 else 0.0 * o_prev.se
                            if np.all(np.isfinite(r.se)) and np.all(r.se >= lb * (1.0 - 1e-9) - 1e-300):
                                se_noise += 1      # (counted apart: not a band, a quantity made of noise on both sides)
                            else:
                                se_ok = False
                        else:
                            se_ok = False
                # 6 iterations at most; a system that converges to machine precision earlier may stop one
                # iteration apart (eps-level tests): x must agree either way
                tol3 = tol_long
                # an eps-level stopping test (1 + test2 <= 1) that fires for one and not the other at the
                # same iteration changes istop but not x: accepted when x agrees to 1e-12
                # (a 2-row system is solved exactly after 2 iterations; the reference runs 2 more on noise)
                ok = e1 < 1e-12 and e2 < 1e-12 and se_ok and (e3 < 1e-12 or (e3 < tol3 and abs(r.itn - o.itn) <= 1 and
                                                                              (r.istop == o.istop or r.itn != o.itn)))
                info = s.info()
                total += 1
                if info.get("pat_wide") or info.get("pat_wide_t"):
                    wide_hits += 1
                if ok and not e3 < TIGHT:
                    widened += 1
                if ok and se_widened:
                    se_wide += 1
                worst = max(worst, e3)
                worst_ref = max(worst_ref, e3_ref)
            except Exception as ex:        # noqa: BLE001
                ok, e1, e2, e3, info = False, -1, -1, -1, repr(ex)
            finally:
                if sh is not None:
                    sh.close()
            if not ok:
                bad += 1
                ri, rn = (r.istop, r.itn) if e1 >= 0 else (None, None)
                print(f"FAIL case {case} m={m} n={n} nnz={irow.size} layout={lay} e1={e1:.2e} e2={e2:.2e} e3={e3:.2e} se={e4:.2e} istop {ri}/{o.istop} itn {rn}/{o.itn} {info}", flush=True)
        if bands:
            print(f"case {case}: worst GPU layout {worst_ref:.2e} from the reference's x, {worst:.2e} from the nearer of the "
                  f"reference's and the accurately summed one's", flush=True)
    for k in KNOBS + ENGINE_KNOBS:
        os.environ.pop(k, None)
    if se_wide > MAX_WIDENED_SHARE * max(total, 1):   # (the standard errors' measured bands are no blanket either)
        bad += 1
        print(f"FAIL: {se_wide} of {total} standard-error results needed a measured band (at most {MAX_WIDENED_SHARE:.0%} may)")
    if verbose:
        print(f"({se_wide} of {total} standard-error vectors were accepted on the reference's own measured band, {se_noise} more "
              f"where the reference's last iteration ran on rounding noise; "
              f"{wide_hits} results went through the wide row-pattern table)")
        print(f"{ncases} cases x {len(ENGINES if engine else LAYOUTS)} {'engine variants' if engine else 'layouts'}: {bad} failures; {widened} of {total} results needed more than "
              f"{TIGHT:g} (at most {MAX_WIDENED_SHARE:.0%} may)")
    return bad, widened, total


EPS32 = float(np.finfo(np.float32).eps)


def run_real32(ncases, seed, verbose=True):
    """The same cases with real32 storage (values, vectors) in every layout.  A REAL32 product is the binary64 sum of exact
    products of real32 numbers, rounded to real32 ONCE (tests/test_gpu_real32.py): held element by element to one real32
    rounding of the oracle's binary64 product of the same real32-valued inputs.  The short solve rounds every vector to
    real32 once per iteration: where the binary64 reference itself is insensitive (tolerance TIGHT, see the header) and
    both run the same number of iterations, x must lie within 2e-3 of the binary64 oracle's."""
    rs = np.random.RandomState(seed)
    po = oracle.port()
    bad = total = compared = 0
    for case in range(ncases):
        m, n, irow, icol, a, b = make_case(rs)
        xp, yp = rs.uniform(-1, 1, size=n), rs.uniform(-1, 1, size=m)
        a32, b32, xp32, yp32 = (v.astype(np.float32) for v in (a, b, xp, yp))
        a64, b64, xp64, yp64 = (v.astype(np.float64) for v in (a32, b32, xp32, yp32))
        _, y_ref = po.aprod(1, m, n, irow, icol, a64, xp64, yp64)
        x_ref, _ = po.aprod(2, m, n, irow, icol, a64, xp64, yp64)
        o = po.solve(m, n, irow, icol, a64, b64, damp=1e-2, itnlim=6)
        longest = max(int(np.bincount(irow - 1, minlength=m).max()), int(np.bincount(icol - 1, minlength=n).max())) if irow.size else 0
        for lay in LAYOUTS:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(lay)
            try:
                s = lsqr_solver_ez().initialize(m, n, a32, irow, icol, itnlim=6, real32=True)
                x, y = xp32.copy(), yp32.copy()
                s.aprod(1, m, n, x, y)
                f1 = EPS32 * np.abs(y_ref) + 1e-12 * max(float(np.max(np.abs(y_ref))), 1.0)
                e1 = float(np.max(np.abs(y.astype(np.float64) - y_ref) / f1)) if m else 0.0
                x, y = xp32.copy(), yp32.copy()
                s.aprod(2, m, n, x, y)
                f2 = EPS32 * np.abs(x_ref) + 1e-12 * max(float(np.max(np.abs(x_ref))), 1.0)
                e2 = float(np.max(np.abs(x.astype(np.float64) - x_ref) / f2)) if n else 0.0
                r = s.solve(b32, 1e-2)
                ok = e1 <= 1.0 and e2 <= 1.0 and np.all(np.isfinite(r.x))
                e3 = -1.0
                if longest <= 16 and o.itn > 0 and r.itn == o.itn:
                    e3 = float(np.linalg.norm(r.x.astype(np.float64) - o.x) / max(np.linalg.norm(o.x), 1e-300))
                    ok = ok and e3 <= 2e-3
                    compared += 1
                total += 1
                info = s.info()
            except Exception as ex:        # noqa: BLE001
                ok, e1, e2, e3, info = False, -1, -1, -1, repr(ex)
            if not ok:
                bad += 1
                print(f"FAIL (REAL32) case {case} m={m} n={n} nnz={irow.size} layout={lay} e1={e1:.2e} e2={e2:.2e} "
                      f"e3={e3:.2e} (in real32 roundings / relative) {info}", flush=True)
    for k in KNOBS:
        os.environ.pop(k, None)
    if verbose:
        print(f"{ncases} cases x {len(LAYOUTS)} layouts in REAL32: {bad} failures; {compared} of {total} solves compared "
              f"with the binary64 oracle's")
    return bad, 0, total


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--only="):
            only = {int(t) for t in a[7:].split(",")}
    bad, widened, total = run(int(args[0]) if args else 60, int(args[1]) if len(args) > 1 else 1,
                              bands="--bands" in sys.argv, only=only, engine="--engine" in sys.argv,
                              real32="--real32" in sys.argv)
    sys.exit(1 if bad or widened > MAX_WIDENED_SHARE * max(total, 1) else 0)
