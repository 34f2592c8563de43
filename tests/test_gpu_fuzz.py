"""Random small systems through every storage layout the build can choose (forced by the ablation
knobs: row and structure patterns, sliced ELL packed / unpacked, row windows, L2 panels with and without
skew handling, LDS panels, column-swept row blocks with small blocks and column splits, 64-bit row pointers,
16- and 32-bit columns) against the oracle: both aprod modes to rounding and a short solve.  Shapes on
purpose: m, n in {1, 2, 63, 64, 65, ...}, empty rows and columns, nnz = 0, duplicates, one very long row,
dictionary and arbitrary values, scrambled COO order.  tests/fuzz_layouts.py holds the generator and the
acceptance rule -- every case is compared, none is exempt: where rows are long the tolerance is a multiple of
what the REFERENCE's own x moves by under a permutation of its input and under one ulp of one norm, measured
per case (`python tests/fuzz_layouts.py 500 7 --bands` runs more and prints the bands)."""
import os

import pytest

import fuzz_layouts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_systems_through_every_layout(seed):
    old = {k: os.environ.get(k) for k in fuzz_layouts.KNOBS}
    try:
        bad, widened, total = fuzz_layouts.run(40, seed, verbose=False)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    print(f"seed {seed}: {widened} of {total} results needed more than {fuzz_layouts.TIGHT:g}")
    assert bad == 0
    assert total >= 40 * len(fuzz_layouts.LAYOUTS) - 5
    assert widened <= fuzz_layouts.MAX_WIDENED_SHARE * total


def test_seed_605_systems_solved_exactly_before_the_last_iteration():
    """Round 5's campaign seed 605, cases 26 (m = 2, n = 65) and 34 (m = 5000, n = 65): TWO nonzeros each.  Such a system
    is solved exactly after two iterations; the third -- which the reference runs as well -- normalises rounding noise into
    a unit vector and the standard errors add (w / rho)^2 of it (src/lsqr.f90:733-737): se differed by 0.38 / 2e-9
    relative in EVERY layout while x, the norms, istop and itn agreed to 1e-15, and the bands the rule measures on the
    reference (permutations, one ulp of one norm) are zero for two nonzeros.  The rule now recognises the case on the
    reference itself (its test2 one iteration before the last is at rounding level: fuzz_layouts
    last_iteration_ran_on_noise) and holds what is left to hold -- se finite and no smaller than what the iterations
    before the noise had accumulated; x, the norms and the counts as for every other case."""
    old = {k: os.environ.get(k) for k in fuzz_layouts.KNOBS}
    try:
        bad, widened, total = fuzz_layouts.run(35, 605, verbose=False, only=(26, 34))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert bad == 0 and total == 2 * len(fuzz_layouts.LAYOUTS)


@pytest.mark.parametrize("seed", [4, 5])
def test_random_systems_through_the_sharded_engine(seed):
    """The same generator through the C++ engine with 2, 3, 5 and 8 ranks on the one device (loopback exchanges; the
    overlapped schedule for two of them; column-swept blocks forced for two): both aprod modes of the sharded handle
    and the short solve, against the oracle, same acceptance rule."""
    keys = fuzz_layouts.KNOBS + fuzz_layouts.ENGINE_KNOBS
    old = {k: os.environ.get(k) for k in keys}
    try:
        bad, widened, total = fuzz_layouts.run(24, seed, verbose=False, engine=True)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    print(f"seed {seed}: {widened} of {total} engine results needed more than {fuzz_layouts.TIGHT:g}")
    assert bad == 0
    assert total >= 24 * len(fuzz_layouts.ENGINES) - 5
    assert widened <= fuzz_layouts.MAX_WIDENED_SHARE * total


@pytest.mark.parametrize("seed", [6, 7])
def test_random_systems_through_every_layout_in_real32(seed):
    """The same generator with REAL32 handles (src/lsqr_kinds.F90:16-17) in every layout: both products element by
    element within ONE real32 rounding of the binary64 oracle's product of the same real32-valued inputs, and the short
    solve within 2e-3 of the binary64 oracle's where that one is insensitive (fuzz_layouts.run_real32)."""
    old = {k: os.environ.get(k) for k in fuzz_layouts.KNOBS}
    try:
        bad, _, total = fuzz_layouts.run(30, seed, verbose=False, real32=True)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert bad == 0
    assert total >= 30 * len(fuzz_layouts.LAYOUTS) - 5
