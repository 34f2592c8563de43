"""The C-ABI library loads and exports every symbol include/lsqrhip.h declares; without a
GPU every compute entry point fails loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from lsqr_amd import capi, problems as P
from lsqr_amd.capi import LsqrHipError
from lsqr_amd.solver import lsqr_solver_ez

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lsqrhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsqrhip_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build()"


def test_every_declared_symbol_is_exported_and_bound():
    L = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(capi.EXPORTS) == syms, "capi.py must bind exactly what the header declares"


def test_error_strings_are_the_references_error_stop_messages():
    L = capi.lib()
    want = {1: "invalid a,icol,irow sizes in initialize_ez",          # src/lsqr.f90:109
            2: "invalid irow or m in initialize_ez",                  # :110
            3: "invalid icol or n in initialize_ez",                  # :111
            4: "lsqr_solver_ez class not properly initialized",       # :152
            5: "invalid mode in aprod_ez"}                            # :197
    for code, msg in want.items():
        assert L.lsqrhip_error_string(code).decode() == msg


def test_timing_struct_layout():
    assert C.sizeof(capi.Timing) == 5 * 8 + 6 * 8 + 8  # 5 doubles, 6 int64, int + padding


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="this check is for boxes without a GPU")
def test_no_device_means_loud_failure_not_fallback():
    assert capi.device_count() == 0
    p = P.readme_3x3()
    with pytest.raises(LsqrHipError) as e:
        lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(LsqrHipError):
        lsqr_solver_ez().solve(np.zeros(3), 0.0)           # not initialised -> code 4, still an error
    # size validation happens before any device work (reference :109)
    with pytest.raises(LsqrHipError) as e:
        lsqr_solver_ez().initialize(3, 3, p.a[:5], p.irow, p.icol)
    assert e.value.code == capi.ERR_SIZES
