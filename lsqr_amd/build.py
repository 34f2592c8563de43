"""In-tree builds: liblsqrhip.so (hipcc, gfx950) and the Fortran host layer (amdflang).

Everything lands under lsqr_amd/lib/ so the binaries travel with the repo snapshot to
the GPU box (they are git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
FSRC = os.path.join(HERE, "fortran")
LIBDIR = os.path.join(HERE, "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FC = os.environ.get("AMDFLANG", "/opt/rocm/bin/amdflang")

# -ffp-contract=off: the fused kernels keep the reference's separate multiply/add
# roundings (DESIGN.md "numerics"); the path is HBM-bound, FMA buys nothing.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-Wall", "-Wno-unused-result"]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "liblsqrhip.so")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                  [os.path.join(ROOT, "include", "lsqrhip.h")])
    if not force and _newer(out, srcs):
        return out
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}")
    cmd = [HIPCC] + HIP_FLAGS + ["-o", out] + sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


def build_fortran(force: bool = False, verbose: bool = False) -> list[str]:
    """Host Fortran modules (lsqr_kinds, lsqpblas_module, lsqr_module) + test drivers."""
    mk = os.path.join(FSRC, "Makefile")
    if not os.path.exists(mk):
        return []
    cmd = ["make", "-C", FSRC, f"FC={FC}"] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return sorted(glob.glob(os.path.join(LIBDIR, "*")))


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_fortran(force, verbose)
