"""The Fortran host layer (lsqr_amd/fortran: modules lsqr_kinds, lsqpblas_module,
lsqr_module with the reference's names and signatures).

* CPU: a user type extending `lsqr_solver` with its own aprod (host path) against the
  reference run on the same matrix; the EZ driver must fail loudly without a GPU.
* GPU: Fortran -> ISO_C_BINDING -> liblsqrhip.so -> HIP kernels against the oracle.
"""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from lsqr_amd import problems as P

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lsqr_amd", "lib")
NUM = r"[-+]?\d\.\d+E[-+]\d+"


def run(exe, check=True, env=None, args=()):
    path = os.path.join(LIB, exe)
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    return subprocess.run([path, *[str(a) for a in args]], capture_output=True, text=True, timeout=300, check=check,
                          env=None if env is None else {**os.environ, **env})


def numbers(line):
    return np.array([float(t) for t in re.findall(NUM, line)])


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_user_aprod_host_path_matches_reference():
    """lsqr / acheck / xcheck of the abstract class (own Fortran code) on a dense user operator,
    compared with the reference algorithm on the same matrix given as COO."""
    out = run("test_user_aprod").stdout
    assert "USER APROD TESTS PASSED" in out and "ACHECK inform= 0" in out
    lines = {l.split("=")[0].strip(): l for l in out.splitlines() if "=" in l}
    m, n, damp = 40, 25, 0.125
    i, j = np.meshgrid(np.arange(1, m + 1), np.arange(1, n + 1), indexing="ij")
    A = 1.0 / (i + 2 * j + 1) + 2.0 * (i == j)
    b = 1.0 / np.arange(1, m + 1) - 0.25
    irow = i.T.reshape(-1).astype(np.int32)   # column-major triplets
    icol = j.T.reshape(-1).astype(np.int32)
    a = A.T.reshape(-1)
    eng = oracle.ref() or oracle.port()
    r = eng.solve(m, n, irow, icol, a, b, damp=damp, atol=1e-12, btol=1e-12, conlim=1e8, itnlim=200, wantse=True)
    istop, itn = [int(t) for t in re.findall(r"=\s*(\d+)", lines["LSQR istop"])]
    assert (istop, itn) == (r.istop, r.itn)
    x = numbers(lines["LSQR x"])
    se = numbers(lines["LSQR se"])
    norms = numbers(lines["LSQR norms"])
    assert np.linalg.norm(x - r.x) <= 1e-10 * np.linalg.norm(r.x)
    assert np.linalg.norm(se - r.se) <= 1e-9 * np.linalg.norm(r.se)
    assert abs(norms[0] - r.anorm) <= 1e-10 * r.anorm
    assert abs(norms[2] - r.rnorm) <= 1e-10 * r.rnorm
    assert abs(norms[4] - r.xnorm) <= 1e-10 * r.xnorm
    assert int(re.findall(r"tests=\s*(\d)", out)[0]) == 3      # damped least-squares solution


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_ez_driver_fails_loudly_without_a_device():
    p = run("test_ez", check=False)
    assert p.returncode != 0
    assert "no usable MI355X" in (p.stdout + p.stderr)
    assert "EZ TESTS PASSED" not in p.stdout


@pytest.mark.gpu
def test_ez_driver_on_gpu_matches_oracle():
    p = run("test_ez", check=False)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = p.stdout
    assert "EZ TESTS PASSED" in out and "OPTIONS OK" in out
    # the reference's iteration log is printed for the toy systems (nout = output_unit)
    assert " Enter LSQR.       Least-squares solution of  Ax = b" in out
    assert "A solution to Ax = b was found, given atol, btol" in out
    toy = [l for l in out.splitlines() if l.startswith("TOY readme_3x3")][0]
    assert "istop= 1" in toy
    assert np.allclose(numbers(toy), [1.2424242424242424, -6.0606060606060594e-02, -4.0404040404040407e-02],
                       rtol=1e-10, atol=0)
    damped = [l for l in out.splitlines() if l.startswith("DAMPED istop")][0]
    assert np.allclose(numbers(damped), [8.10640473795055549e-01, -2.86387641921484436e-02, -1.64648479828245174e-02],
                       rtol=1e-10, atol=0)        # SURVEY.md 8c 'T1 damped' golden
    # generated Poisson system: Fortran host -> C-ABI -> HIP against the CPU oracle
    prob = P.poisson2d(64, 64)
    head = [l for l in out.splitlines() if l.startswith("POISSON nx=")][0]
    assert f"nnz={prob.nnz}" in head and "istop=5" in head and "itn=60" in head
    o = oracle.port().solve(prob.m, prob.n, prob.irow, prob.icol, prob.a, prob.b, itnlim=60)
    nrm = numbers([l for l in out.splitlines() if l.startswith("POISSON anorm")][0])
    xs = numbers([l for l in out.splitlines() if l.startswith("POISSON x(1)")][0])
    assert abs(nrm[0] - o.anorm) <= 1e-10 * o.anorm
    assert abs(nrm[1] - o.rnorm) <= 1e-10 * o.rnorm
    assert abs(nrm[2] - o.xnorm) <= 1e-10 * o.xnorm
    want = np.array([o.x[0], o.x[prob.n // 2 - 1], o.x[-1]])
    assert np.max(np.abs(xs - want)) <= 1e-10 * np.linalg.norm(o.x, np.inf)


@pytest.mark.gpu
@pytest.mark.parametrize("ngpu", [1, 3])
def test_sharded_initialize_from_fortran(ngpu):
    """`initialize(..., ngpu=N)`: Fortran -> lsqrhip_create_sharded -> the C++ RCCL engine.  At ngpu = 1 on
    a one-GPU box against the oracle (and at ngpu = 3 with the exchanges looped back inside the process,
    LSQRHIP_SHARD_LOOPBACK=1: three ranks on the one device); asking for more GPUs than the node has must
    `error stop`."""
    path = os.path.join(LIB, "test_sharded")
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    env = {**os.environ, "LSQRHIP_SHARD_LOOPBACK": "1" if ngpu > 1 else "0"}
    p = subprocess.run([path, str(ngpu)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = p.stdout
    assert f"SHARDED TESTS PASSED ngpu={ngpu}" in out
    readme = [l for l in out.splitlines() if l.startswith("README istop")][0]
    assert "istop= 1" in readme
    assert np.allclose(numbers(readme), [1.2424242424242424, -6.0606060606060594e-02, -4.0404040404040407e-02],
                       rtol=1e-10, atol=0)
    prob = P.poisson2d(64, 64)
    o = oracle.port().solve(prob.m, prob.n, prob.irow, prob.icol, prob.a, prob.b, itnlim=60)
    assert "istop=5 itn=60" in [l for l in out.splitlines() if l.startswith("POISSON nx=")][0]
    nrm = numbers([l for l in out.splitlines() if l.startswith("POISSON_NORMS")][0])
    xs = numbers([l for l in out.splitlines() if l.startswith("POISSON_X")][0])
    assert abs(nrm[0] - o.anorm) <= 1e-10 * o.anorm and abs(nrm[1] - o.rnorm) <= 1e-10 * o.rnorm
    n = prob.n
    want = np.array([o.x[0], o.x[n // 3 - 1], o.x[n // 2 - 1], o.x[-1]])
    assert np.max(np.abs(xs - want)) <= 1e-10 * np.linalg.norm(o.x, np.inf)
    import torch
    too_many = torch.cuda.device_count() + 1
    env["LSQRHIP_SHARD_LOOPBACK"] = "0"
    p = subprocess.run([path, str(too_many)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "SHARDED TESTS PASSED" not in p.stdout
    assert "no usable MI355X" in (p.stdout + p.stderr) and f"ngpu = {too_many}" in (p.stdout + p.stderr)


REF_MESSAGES = {   # the reference's `error stop` strings, src/lsqr.f90:109-111, 152, 197
    "sizes": "invalid a,icol,irow sizes in initialize_ez",
    "irow": "invalid irow or m in initialize_ez",
    "icol": "invalid icol or n in initialize_ez",
    "notinit": "lsqr_solver_ez class not properly initialized",
    "dims": "lsqr_solver_ez class not properly initialized",
    "mode": "invalid mode in aprod_ez",
}


def run_err(which):
    path = os.path.join(LIB, "test_errors")
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    return subprocess.run([path, which], capture_output=True, text=True, timeout=120)


@pytest.mark.parametrize("which", ["sizes", "notinit"])
def test_fortran_error_stops_that_need_no_device(which):
    """Checked before any device work, exactly like the reference checks them first."""
    p = run_err(which)
    assert p.returncode != 0
    assert REF_MESSAGES[which] in (p.stdout + p.stderr)
    assert "NO ERROR RAISED" not in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["irow", "icol", "dims", "mode"])
def test_fortran_error_stops_on_gpu(which):
    p = run_err(which)
    assert p.returncode != 0
    assert REF_MESSAGES[which] in (p.stdout + p.stderr)
    assert "NO ERROR RAISED" not in p.stdout


@pytest.mark.gpu
def test_fortran_error_driver_ok_case():
    p = run_err("ok")
    assert p.returncode == 0 and "OK istop= 1" in p.stdout


@pytest.mark.gpu
def test_fortran_device_operator_and_user_subclass_on_gpu():
    """lsqr_device_module: the reference's test problem P(2000,1000,40,3,1e-9) as a device
    operator driven from Fortran with the reference's argument lists, and a user type extending
    lsqr_solver_device whose aprod_device enqueues library work on the stream it is handed."""
    out = run("test_device_operator").stdout
    assert "DEVICE OPERATOR TESTS PASSED" in out
    line = {k: l for l in out.splitlines() for k in ("LSTP acond", "ACHECK", "LSQR istop", "LSQR norms", "XCHECK",
                                                      "ENORM", "X8", "USER istop", "USER maxdiff") if l.startswith(k)}
    o = oracle.port().lstp_test(2000, 1000, 40, 3, 1e-9)
    acond, rnorm = numbers(line["LSTP acond"])
    assert acond == pytest.approx(o["acond_lstp"], rel=1e-14) and rnorm == pytest.approx(o["rnorm_lstp"], rel=1e-13)
    assert "inform= 0" in line["ACHECK"]
    istop, itn = (int(t) for t in re.findall(r"=\s*(\d+)", line["LSQR istop"]))
    assert istop == o["istop"] == 3
    # the measured spread of this problem's iteration count under legal summation orders
    # (tests/golden/lstp_itn_band.json, see tests/test_gpu_operator.py)
    import json
    band = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lstp_itn_band.json")))
    itns = list(next(b for b in band if (b["m"], b["n"], b["npower"]) == (2000, 1000, 3))["itn"].values())
    width = max(3, max(itns) - min(itns))
    assert itns[0] == o["itn"] and min(itns) - width <= itn <= max(itns) + width
    assert int(re.search(r"inform,tests=\s*(\d+)", line["XCHECK"]).group(1)) == o["xcheck_inform"]
    enorm = numbers(line["ENORM"])[0]
    assert enorm <= 50 * o["enorm"] + 1e-13
    np.testing.assert_allclose(numbers(line["X8"]), o["xtrue"][:8], rtol=0,
                               atol=1.01 * enorm * (1.0 + np.linalg.norm(o["xtrue"])))
    uistop, uitn, calls = (int(t) for t in re.findall(r"=\s*(\d+)", line["USER istop"]))
    assert (uistop, uitn) == (istop, itn) and calls >= 2 * itn + 1
    assert numbers(line["USER maxdiff"])[0] == 0.0


@pytest.mark.gpu
def test_fortran_device_operator_in_the_real32_build_on_gpu():
    """lsqr_device_module compiled -DREAL32 binds the REAL32 operator entry points (lsqrhip_create_operator_f32,
    lsqrhip_lstp_create_f32, lsqrhip_solve_f32, lsqrhip_acheck_f32, lsqrhip_xcheck_f32): real32 vectors on the device.
    The suite's first problem, acheck -> lsqr -> xcheck -> error, and the user subclass, against what the UNMODIFIED
    reference compiled -DREAL32 reports for it (tests/golden/real32_lstp_ref.json)."""
    import json
    out = run("test_device_operator32").stdout
    assert "DEVICE OPERATOR TESTS PASSED" in out
    line = {k: l for l in out.splitlines() for k in ("ACHECK", "LSQR istop", "XCHECK", "ENORM", "USER istop",
                                                      "USER maxdiff") if l.startswith(k)}
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real32_lstp_ref.json")))
    g = next(p for p in ref if (p["m"], p["n"], p["npower"]) == (2000, 1000, 2))
    assert "inform= 0" in line["ACHECK"] and g["acheck_ok"]
    istop, itn = (int(t) for t in re.findall(r"=\s*(\d+)", line["LSQR istop"]))
    assert istop == g["istop"] and 0.5 * g["itn"] <= itn <= 1.05 * g["itn"]
    assert int(re.search(r"inform,tests=\s*(\d+)", line["XCHECK"]).group(1)) == g["xcheck_inform"]
    assert numbers(line["ENORM"])[0] <= 1e-3 and g["success"]
    uistop, uitn, calls = (int(t) for t in re.findall(r"=\s*(\d+)", line["USER istop"]))
    assert (uistop, uitn) == (istop, itn) and calls >= 2 * itn + 1 and numbers(line["USER maxdiff"])[0] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["real32", "mixed", "real32 sharded over 3 ranks", "mixed sharded over 3 ranks"])
def test_real32_build_on_gpu(mode):
    """-DREAL32 build of the host layer (wp = real32 like the reference's REAL32 macro).
    "real32" (the default): real32 arrays in, out AND on the device (values, u, v, w, x, se), binary64
    in registers.  "mixed" (LSQRHIP_REAL32_MIXED=1): binary64 on the device, real32 at the boundary.
    Against (1) the binary64 oracle on the same real32-valued inputs: mixed agrees to real32 rounding of
    the outputs, all-real32 to what rounding the vectors once per iteration costs; (2) the unmodified
    reference compiled with -DREAL32 (tests/golden/real32_ref.json): agreement to what an all-real32
    iteration can hold, and never further from the binary64 answer than that reference is."""
    import json
    # "... sharded": initialize(..., ngpu=3) of the same build (lsqrhip_create_sharded_f32; three ranks on this GPU
    # through the loopback exchanges): real32 storage in every row block and on the links
    sharded = "sharded" in mode
    mode = mode.split()[0]
    out = run("test_real32", env={"LSQRHIP_REAL32_MIXED": "1" if mode == "mixed" else "0",
                                  "LSQRHIP_SHARD_LOOPBACK": "1" if sharded else "0"}, args=(3,) if sharded else ()).stdout
    assert "REAL32 TESTS PASSED" in out
    line = {l.split("=")[0].strip(): l for l in out.splitlines() if "=" in l}
    ref32 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real32_ref.json")))
    # the same stencil in binary64 (every coefficient and b is exactly representable in real32)
    nx, ny = 40, 30
    irow, icol, a, b = [], [], [], []
    for j in range(1, ny + 1):
        for i in range(1, nx + 1):
            k = (j - 1) * nx + i
            ent = [(k, 4.25)]
            if i > 1: ent.append((k - 1, -1.125))
            if i < nx: ent.append((k + 1, -0.875))
            if j > 1: ent.append((k - nx, -1.0625))
            if j < ny: ent.append((k + nx, -0.9375))
            for c, v in ent:
                irow.append(k); icol.append(c); a.append(v)
            b.append((7 * k % 13) * 0.25 - 1.5)
    n = nx * ny
    f32 = lambda v: float(np.float32(v))
    o = oracle.port().solve(n, n, irow, icol, a, np.array(b), damp=0.0625, atol=f32(1e-7), btol=f32(1e-7),
                            itnlim=500, wantse=True)
    istop, itn = (int(t) for t in re.findall(r"=\s*(\d+)", line["STENCIL32 istop"]))
    x = numbers(line["STENCIL32 x"])
    norms = numbers(line["STENCIL32 norms"])
    x32 = np.array(ref32["STENCIL32 x"]["nums"])
    if mode == "mixed":
        assert (istop, itn) == (o.istop, o.itn)
        assert np.max(np.abs(x - o.x)) <= 1.2e-7 * np.max(np.abs(o.x))      # real32 rounding of the output
        np.testing.assert_allclose(norms[[2, 4]], [o.rnorm, o.xnorm], rtol=1.2e-7)
    else:
        assert istop == o.istop and abs(itn - o.itn) <= 3
        assert np.linalg.norm(x - o.x) <= 1e-4 * np.linalg.norm(o.x)
        assert np.linalg.norm(x - o.x) <= 1.5 * np.linalg.norm(x32 - o.x)  # no worse than the all-real32 reference
        np.testing.assert_allclose(norms[[2, 4]], [o.rnorm, o.xnorm], rtol=1e-4)
    # anorm / acond are running sums over all 170 Lanczos steps: two binary64 runs that differ in
    # summation order drift apart in them long before they do in x
    np.testing.assert_allclose(norms[[0, 1]], [o.anorm, o.acond], rtol=5e-3)
    np.testing.assert_allclose(numbers(line["STENCIL32 se"]), o.se[:8], rtol=1e-2 if mode == "mixed" else 3e-2)
    # the all-real32 reference: same stopping reason, same answer to ~1e-4 (it needs 171
    # iterations where binary64 arithmetic needs fewer: its Lanczos vectors lose orthogonality sooner)
    r = ref32["STENCIL32 istop"]["ints"]
    assert r[0] == istop and itn <= r[1] + 3
    assert np.linalg.norm(x - x32) <= 2e-4 * np.linalg.norm(x32)
    assert np.allclose(numbers(line["README32 istop,x"])[-3:], ref32["README32 istop,x"]["nums"][-3:], rtol=2e-5)
