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
