"""The ORDER of memory requests in the column-swept product's lock-step loop (csrc/csb.h "LOCK STEP"), held on the ISA
hipcc emits for gfx950 (CPU test: hipcc cross-compiles without a GPU).

The loop exists for one property: inside a step every wave requests ALL its gathers of x, then passes the barrier, then
requests the next step's (value, index) stream -- never a gather behind a new stream request (the CU's vector L1 returns
data in request order across its waves).  The first build of round 5 lost it silently: with a branch around the adds of a
wave's last, partial step the compiler sank that chunk's gathers into the branch -- behind the barrier and the stream
requests -- and config 4 ran at 2.82 instead of 2.42 ms.  Nothing functional notices (the sums are integer sums), so this
test reads the assembly."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not (shutil.which(HIPCC) or os.path.exists(HIPCC)), reason="needs hipcc")


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "lsqrhip_dev.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", out,
           os.path.join(ROOT, "lsqr_amd", "csrc", "lsqrhip.hip")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return open(out).read()


def kernel_body(asm, mangled_prefix):
    m = re.search(r"^(%s\w*):" % re.escape(mangled_prefix), asm, re.M)
    assert m, mangled_prefix
    end = asm.index("s_endpgm", m.end())
    return asm[m.end():end]


def tokens(body):
    """G = a gather of x (a plain 8- or 4-byte global load through a vector address), N = a non-temporal stream load,
    B = s_barrier; everything else dropped."""
    out = []
    for line in body.splitlines():
        t = line.strip()
        if t.startswith("s_barrier"):
            out.append("B")
        elif t.startswith("global_load_dword"):
            if " nt" in t:
                out.append("N")
            elif ", off" in t and "v[" in t.split(",")[1]:
                out.append("G")
    return "".join(out)


@pytest.mark.parametrize("vt,narrow", [("d", 0), ("d", 1), ("f", 0)])
@pytest.mark.parametrize("K", [1, 2])
def test_gathers_of_a_step_precede_the_barrier_and_the_stream_follows(device_asm, vt, narrow, K):
    body = kernel_body(device_asm, "_ZN7lsqrhip10k_spmv_csbI%sLb%dELi%dE" % (vt, narrow, K))
    t = tokens(body)
    stream = (8 if not narrow else 6) * K          # wide: 4 index + 4 value loads per chunk; narrow: 2 (rows, deltas) + 4
    pat = "G{%d}BN{%d,}" % (4 * K, stream)
    hits = re.findall(pat, t)
    # the steady-state loop is unrolled by two (two register sets): at least two such steps, each with ALL 4 K gathers in
    # front of the barrier
    assert len(hits) >= 2, (pat, t[-400:])
    # ... and nowhere a barrier with stream requests right behind it and gathers after those (the sunk form)
    assert not re.search(r"BN+G", t), t[-400:]


def test_free_running_form_has_no_barrier_in_its_loop(device_asm):
    body = kernel_body(device_asm, "_ZN7lsqrhip10k_spmv_csbIdLb0ELi0E")
    t = tokens(body)
    assert "GGGGN" in t and not re.search(r"G{4}BN", t)
