"""The drop-in proof: the reference's OWN test programs, compiled UNCHANGED where they lie
under /root/reference against this repository's Fortran host layer (lsqr_amd/lib/fmod + liblsqr_amd_f.a +
liblsqrhip.so), exactly as a user of the reference would switch libraries.

* test/lsqrtest_module.f90 + test/lsqrtest.f90 (a user type that extends `lsqr_solver` and overrides
  `aprod`, :35-44; 18 problems through lsqr / acheck / xcheck): runs on the HOST path, needs no GPU;
  every problem must stop with istop = 3 and the success pattern must be the one the reference ships in
  test/LSQR.LIS (16 "successful", 2 "failed": problems 5 and 6) -- tests/golden/LSQR_shipped_facts.json.
* test/lsqrtest_ez.f90 (`lsqr_solver_ez`, :18-104): must compile and link; without a GPU it must stop with
  the no-device message (there is no CPU fallback), on a GPU it must pass the program's own checks.

Nothing of the reference is copied: it is compiled in place, like oracle/Makefile does.  The build
container only -- /root/reference does not exist on the GPU box, where these tests skip."""
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lsqr_amd", "lib")
REF = "/root/reference"
FC = os.environ.get("AMDFLANG", "/opt/rocm/bin/amdflang")

pytestmark = pytest.mark.skipif(
    not (os.path.isdir(os.path.join(REF, "test")) and os.path.exists(FC)),
    reason="needs the reference sources under /root/reference and amdflang (build container only)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def link(tmp, name, sources):
    for f in ("liblsqr_amd_f.a", "liblsqrhip.so", "fmod/lsqr_module.mod"):
        assert os.path.exists(os.path.join(LIB, f)), f"{f} missing: run __graft_entry__.build()"
    exe = os.path.join(tmp, name)
    cmd = [FC, "-O2", "-I" + os.path.join(LIB, "fmod"), "-J" + str(tmp)] + [os.path.join(REF, "test", s) for s in sources] + \
          [os.path.join(LIB, "liblsqr_amd_f.a"), "-L" + LIB, "-llsqrhip", "-Wl,-rpath," + LIB, "-o", exe]
    p = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and "-J" in p.stderr:      # (a flang without -J: module files land in cwd anyway)
        cmd = [c for c in cmd if not c.startswith("-J")]
        p = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, "the reference's program does not compile against the host layer:\n" + p.stderr[-3000:]
    return exe


def test_lsqrtest_18_problems_compile_unchanged_and_run_on_the_host_path(tmp_path):
    exe = link(str(tmp_path), "lsqrtest", ["lsqrtest_module.f90", "lsqrtest.f90"])
    p = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lis = open(os.path.join(str(tmp_path), "LSQR.LIS")).read()
    facts = json.load(open(os.path.join(ROOT, "tests", "golden", "LSQR_shipped_facts.json")))
    assert len(facts) == 18
    istops = [int(t) for t in re.findall(r"istop\s*=\s*(\d+)", lis)]
    assert istops == [f["istop"] for f in facts] == [3] * 18
    verdicts = re.findall(r"LSQR\s+appears to (be successful|have failed)", lis)
    assert len(verdicts) == 18
    ours = [v == "be successful" for v in verdicts]
    # the pattern the reference ships (test/LSQR.LIS:497, 605): only problems 5 and 6 "fail"
    assert [i + 1 for i, ok in enumerate(ours) if not ok] == [5, 6]
    # acheck finds the user's aprod consistent in every problem (its inform = 0 line, src/lsqr.f90:985-994)
    assert len(re.findall(r"aprod seems OK", lis)) == 18
    itns = [int(t) for t in re.findall(r"itn\s*=\s*(\d+)", lis)]
    assert len(itns) == 18
    # Round 5: the host dnrm2 is the reference's one-pass recurrence (lsqr_amd/fortran/lsqrblas.f90), so this path is the
    # reference's arithmetic operation for operation.  The compiled reference sits next to us (oracle/_ref/lsqrtest, the
    # unmodified sources by the same compiler, oracle/Makefile:49-52): the two LSQR.LIS files -- every iteration line of
    # all 18 problems, every norm to its last printed digit -- must be THE SAME TEXT.
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "lsqrtest")
    assert os.path.exists(ref_exe), "oracle/_ref/lsqrtest missing: make -C oracle ref"
    refdir = os.path.join(str(tmp_path), "ref")
    os.makedirs(refdir)
    q = subprocess.run([ref_exe], cwd=refdir, capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    ref_lis = open(os.path.join(refdir, "LSQR.LIS")).read()
    ref_itns = [int(t) for t in re.findall(r"itn\s*=\s*(\d+)", ref_lis)]
    assert itns == ref_itns
    if lis != ref_lis:
        a, b = lis.splitlines(), ref_lis.splitlines()
        bad = [(i + 1, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y][:5]
        raise AssertionError(f"LSQR.LIS differs from the compiled reference's ({len(a)} / {len(b)} lines); first differences: {bad}")
    # (against the log the reference SHIPS -- another compiler's -- the counts agree within that compiler spread)
    for got, f in zip(itns, facts):
        assert abs(got - f["itn"]) <= max(31, f["itn"] // 10), (got, f["itn"])


def test_lsqrtest_ez_compiles_unchanged_and_links(tmp_path):
    exe = link(str(tmp_path), "lsqrtest_ez", ["lsqrtest_ez.f90"])
    p = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    if has_gpu():
        assert p.returncode == 0, out[-2000:]
        assert "FAILED" not in out.upper()
    else:
        assert p.returncode != 0
        assert "no usable MI355X" in out


def test_host_layer_under_address_sanitizer(tmp_path):
    """The Fortran host layer (lsqr_kinds, lsqpblas_module, lsqr_module with its host `lsqr` / `acheck` / `xcheck`, and
    lsqr_device_module) compiled with -fsanitize=address and driven by the reference's 18-problem program on the host
    path: no read or write outside an array, no use after free, nothing leaked (sanitizers exist for CPU code only on
    this pool; the HIP side is held by the bitwise-repeatability tests)."""
    tmp = str(tmp_path)
    src = os.path.join(ROOT, "lsqr_amd", "fortran")
    san = ["-O1", "-g", "-fsanitize=address", "-ffp-contract=off"]
    for f, cpp in (("lsqr_kinds.F90", True), ("lsqrblas.f90", False), ("lsqr_module.f90", False),
                   ("lsqr_device_module.F90", True)):
        p = subprocess.run([FC] + san + (["-cpp"] if cpp else []) + ["-c", os.path.join(src, f)], cwd=tmp,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
    objs = ["lsqr_kinds.o", "lsqrblas.o", "lsqr_module.o", "lsqr_device_module.o"]
    exe = os.path.join(tmp, "lsqrtest_san")
    p = subprocess.run([FC] + san + [os.path.join(REF, "test", "lsqrtest_module.f90"), os.path.join(REF, "test", "lsqrtest.f90")]
                       + objs + ["-L" + LIB, "-llsqrhip", "-Wl,-rpath," + LIB, "-o", exe], cwd=tmp, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1")
    p = subprocess.run([exe], cwd=tmp, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr, p.stderr[-3000:]
    lis = open(os.path.join(tmp, "LSQR.LIS")).read()
    assert len(re.findall(r"LSQR\s+appears to be successful", lis)) == 16


def test_real128_build_of_the_host_layer_is_the_reference_in_binary128(tmp_path):
    """`-DREAL128` (src/lsqr_kinds.F90:20-21; round 5): the host path -- `lsqr_solver` with a user `aprod`, `lsqr`, `acheck`,
    `xcheck`, `lsqpblas_module` -- compiled in binary128 (lsqr_amd/lib/fmod128 + liblsqr_amd_f128.a).  The reference's
    18-problem program, compiled unchanged against it, must write the SAME LSQR.LIS, byte for byte, as the reference's own
    -DREAL128 build (oracle/_ref/lsqrtest128, oracle/Makefile) -- and in binary128 all 18 problems "appear to be
    successful" (the REAL64 build fails problems 5 and 6 by rounding)."""
    tmp = str(tmp_path)
    for f in ("liblsqr_amd_f128.a", "fmod128/lsqr_module.mod"):
        assert os.path.exists(os.path.join(LIB, f)), f"{f} missing: run __graft_entry__.build()"
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "lsqrtest128")
    assert os.path.exists(ref_exe), "oracle/_ref/lsqrtest128 missing: make -C oracle ref"
    exe = os.path.join(tmp, "lsqrtest128")
    cmd = [FC, "-O2", "-I" + os.path.join(LIB, "fmod128"), os.path.join(REF, "test", "lsqrtest_module.f90"),
           os.path.join(REF, "test", "lsqrtest.f90"), os.path.join(LIB, "liblsqr_amd_f128.a"), "-L" + LIB, "-llsqrhip",
           "-Wl,-rpath," + LIB, "-o", exe]
    p = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    p = subprocess.run([exe], cwd=tmp, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lis = open(os.path.join(tmp, "LSQR.LIS")).read()
    refdir = os.path.join(tmp, "ref")
    os.makedirs(refdir)
    q = subprocess.run([ref_exe], cwd=refdir, capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    ref_lis = open(os.path.join(refdir, "LSQR.LIS")).read()
    assert lis == ref_lis
    assert len(re.findall(r"LSQR\s+appears to be successful", lis)) == 18
    assert [int(t) for t in re.findall(r"istop\s*=\s*(\d+)", lis)] == [3] * 18


def test_real128_ez_type_stops_with_a_message(tmp_path):
    """The REAL128 build has no device path: `lsqr_solver_ez%initialize` says so instead of computing in binary64."""
    tmp = str(tmp_path)
    src = os.path.join(tmp, "ez128.f90")
    with open(src, "w") as f:
        f.write("program ez128\n use lsqr_kinds\n use lsqr_module\n implicit none\n type(lsqr_solver_ez) :: s\n"
                " real(wp) :: a(1) = [1.0_wp]\n integer :: ir(1) = [1], ic(1) = [1]\n"
                " call s%initialize(1, 1, a, ir, ic)\n print *, 'not reached'\nend program\n")
    exe = os.path.join(tmp, "ez128")
    p = subprocess.run([FC, "-O2", "-I" + os.path.join(LIB, "fmod128"), src, os.path.join(LIB, "liblsqr_amd_f128.a"),
                        "-L" + LIB, "-llsqrhip", "-Wl,-rpath," + LIB, "-o", exe], cwd=tmp, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    p = subprocess.run([exe], cwd=tmp, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "REAL128 build has no device path" in (p.stdout + p.stderr)
    assert "not reached" not in p.stdout
