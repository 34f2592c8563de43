"""Random small systems through every storage layout the build can choose (forced by the ablation
knobs: sliced ELL packed / unpacked, row windows, L2 panels with and without skew handling, LDS
panels, 64-bit row pointers, 16- and 32-bit columns) against the oracle: both aprod modes to
rounding and a short solve.  Shapes on purpose: m, n in {1, 2, 63, 64, 65, ...}, empty rows and
columns, nnz = 0, duplicates, one very long row, dictionary and arbitrary values, scrambled COO order
(scripts/fuzz_layouts.py holds the generator; `python scripts/fuzz_layouts.py 500 7` runs more)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_HERE = os.path.dirname(os.path.abspath(__file__))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_layouts", os.path.join(_HERE, "..", "scripts", "fuzz_layouts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_systems_through_every_layout(seed):
    old = {k: os.environ.get(k) for k in _fuzz().KNOBS}
    try:
        assert _fuzz().run(40, seed, verbose=False) == 0
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
