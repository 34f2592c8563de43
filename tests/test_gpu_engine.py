"""The C++ sharded engine (csrc/shard_engine.h) at the sizes one GPU allows: the single-process form
lsqrhip_create_sharded(ngpu = 1) behind the ordinary lsqrhip_solve / lsqrhip_aprod, and the
one-rank-per-process form (comm_init + shard_solve, world = 1).  With one rank the exchanges are
local copies, but the stages, the column-slice bookkeeping and the |t3| sqrt(sum w^2) form of dknorm
are those of every world size (tests/test_dist.py runs the same stages under gloo with 2-4 ranks)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from cases import build_cases
from lsqr_amd import capi
from lsqr_amd import problems as P
from lsqr_amd.capi import check, lib
from lsqr_amd.dist import EngineSolver
from lsqr_amd.solver import lsqr_solver_ez

pytestmark = pytest.mark.gpu
CASES = build_cases()


def sharded_handle(p, ngpu):
    h = C.c_void_p()
    irow = np.ascontiguousarray(p.irow, np.int32)
    icol = np.ascontiguousarray(p.icol, np.int32)
    a = np.ascontiguousarray(p.a, np.float64)
    check(lib().lsqrhip_create_sharded(p.m, p.n, a.size, irow.ctypes.data, icol.ctypes.data, a.ctypes.data, ngpu,
                                       C.byref(h)))
    return h


def solve_handle(h, p, o):
    x, se = np.zeros(max(p.n, 1)), np.zeros(max(p.n, 1))
    istop, itn = C.c_int(), C.c_int()
    sc = [C.c_double() for _ in range(5)]
    b = np.ascontiguousarray(p.b, np.float64)
    check(lib().lsqrhip_solve(h, b.ctypes.data, o["damp"], o["atol"], o["btol"], o["conlim"], o["itnlim"],
                              int(o["wantse"]), 0, x.ctypes.data, se.ctypes.data if o["wantse"] else None,
                              C.addressof(istop), C.addressof(itn), *[C.addressof(s) for s in sc]))
    return x[:p.n], se[:p.n], istop.value, itn.value, [s.value for s in sc]


@pytest.mark.parametrize("name", ["random_over_damped", "random_over_se", "poisson_20x20_it50", "shuffled_dups",
                                  "t1_readme_damped", "b_zero", "zero_matrix", "one_by_one", "itnlim_1"])
def test_single_process_sharded_handle_matches_oracle(name):
    p, o = CASES[name]
    h = sharded_handle(p, 1)
    try:
        x, se, istop, itn, sc = solve_handle(h, p, o)
        g = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
        assert istop == g.istop
        if name != "t1_readme_damped":
            assert itn == g.itn
        nx = np.linalg.norm(g.x)
        assert (np.linalg.norm(x - g.x) <= 1e-10 * nx) if nx > 0 else not x.any()
        # (t1_readme_damped: a 3 x 3 system is exhausted after 3 steps; the 4th runs on rounding noise, and so
        # does its contribution to anorm -- tests/golden records the reference's own spread there)
        if g.itn > 0 and itn == g.itn and name != "t1_readme_damped":
            assert abs(sc[0] - g.anorm) <= 1e-10 * g.anorm and abs(sc[2] - g.rnorm) <= 1e-10 * max(g.rnorm, 1e-300) \
                or g.rnorm <= 1e-13 * np.linalg.norm(p.b)
            if o["wantse"]:
                assert np.linalg.norm(se - g.se) <= 1e-9 * np.linalg.norm(g.se)
        # aprod on the sharded handle (host vectors)
        if p.nnz:
            xp, yp = np.linspace(-1, 1, p.n), np.linspace(1, 2, p.m)
            xx, yy = xp.copy(), yp.copy()
            check(lib().lsqrhip_aprod(h, 1, xx.ctypes.data, yy.ctypes.data))
            _, y_ref = oracle.port().aprod(1, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
            assert np.max(np.abs(yy - y_ref)) <= 1e-13 * max(np.max(np.abs(y_ref)), 1.0)
            xx, yy = xp.copy(), yp.copy()
            check(lib().lsqrhip_aprod(h, 2, xx.ctypes.data, yy.ctypes.data))
            x_ref, _ = oracle.port().aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
            assert np.max(np.abs(xx - x_ref)) <= 1e-13 * max(np.max(np.abs(x_ref)), 1.0)
        d = (C.c_int64 * 16)()
        check(lib().lsqrhip_info(h, d))
        assert (d[0], d[1], d[2]) == (p.m, p.n, p.nnz)
    finally:
        check(lib().lsqrhip_destroy(h))


@pytest.fixture
def loopback():
    import os
    old = os.environ.get("LSQRHIP_SHARD_LOOPBACK")
    os.environ["LSQRHIP_SHARD_LOOPBACK"] = "1"
    yield
    os.environ.pop("LSQRHIP_SHARD_LOOPBACK", None)
    if old is not None:
        os.environ["LSQRHIP_SHARD_LOOPBACK"] = old


@pytest.mark.parametrize("ngpu", [2, 3, 8])
@pytest.mark.parametrize("name", ["random_over_damped", "random_over_se", "random_under", "poisson_20x20_it50",
                                  "shuffled_dups", "empty_rows_cols_it20", "powerlaw_small_it10", "b_zero", "zero_matrix",
                                  "one_by_one", "itnlim_1"])
def test_engine_with_several_ranks_on_one_gpu(loopback, name, ngpu):
    """The whole C++ engine at world sizes 2, 3 and 8 on ONE device (LSQRHIP_SHARD_LOOPBACK=1: the three
    exchanges become device copies between the ranks' buffers -- RCCL refuses ranks that share a GPU).  Stages,
    row blocks, column slices (ragged: n not a multiple of the world; more ranks than rows), the rank-ordered
    sums and the replicated scalar recurrences are the production code."""
    p, o = CASES[name]
    h = sharded_handle(p, ngpu)
    try:
        x, se, istop, itn, sc = solve_handle(h, p, o)
        g = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
        assert (istop, itn) == (g.istop, g.itn)
        nx = np.linalg.norm(g.x)
        assert (np.linalg.norm(x - g.x) <= 1e-10 * nx) if nx > 0 else not x.any()
        if g.itn > 0:
            assert abs(sc[0] - g.anorm) <= 1e-10 * g.anorm
            assert abs(sc[2] - g.rnorm) <= 1e-10 * g.rnorm or g.rnorm <= 1e-13 * np.linalg.norm(p.b)
            assert abs(sc[4] - g.xnorm) <= 1e-10 * g.xnorm
            if o["wantse"]:
                assert np.linalg.norm(se - g.se) <= 1e-9 * np.linalg.norm(g.se)
        x2, _, istop2, itn2, sc2 = solve_handle(h, p, o)       # repeats itself bit for bit
        assert np.array_equal(x2, x) and (istop2, itn2, sc2) == (istop, itn, sc)
        if p.nnz:
            xp, yp = np.linspace(-1, 1, p.n), np.linspace(1, 2, p.m)
            xx, yy = xp.copy(), yp.copy()
            check(lib().lsqrhip_aprod(h, 2, xx.ctypes.data, yy.ctypes.data))
            x_ref, _ = oracle.port().aprod(2, p.m, p.n, p.irow, p.icol, p.a, xp, yp)
            assert np.max(np.abs(xx - x_ref)) <= 1e-13 * max(np.max(np.abs(x_ref)), 1.0)
    finally:
        check(lib().lsqrhip_destroy(h))


def test_loopback_result_does_not_depend_on_the_world_size_beyond_rounding(loopback):
    p, o = CASES["poisson_20x20_it50"]
    xs = []
    for ngpu in (1, 2, 5):
        h = sharded_handle(p, ngpu)
        try:
            xs.append(solve_handle(h, p, o))
        finally:
            check(lib().lsqrhip_destroy(h))
    for x, _, istop, itn, sc in xs[1:]:
        assert (istop, itn) == xs[0][2:4]
        assert np.linalg.norm(x - xs[0][0]) <= 1e-11 * np.linalg.norm(xs[0][0])


def test_more_gpus_than_the_node_has_fails_loudly():
    p, _ = CASES["random_over_damped"]
    have = capi.device_count()
    h = C.c_void_p()
    rc = lib().lsqrhip_create_sharded(p.m, p.n, p.a.size, p.irow.ctypes.data, p.icol.ctypes.data, p.a.ctypes.data,
                                      have + 1, C.byref(h))
    assert rc == capi.ERR_NO_DEVICE and not h.value
    assert b"usable gfx950" in lib().lsqrhip_last_error()
    # the reference's validation comes first (src/lsqr.f90:110-111)
    bad = p.irow.copy()
    bad[3] = p.m + 1
    rc = lib().lsqrhip_create_sharded(p.m, p.n, p.a.size, bad.ctypes.data, p.icol.ctypes.data, p.a.ctypes.data, 1,
                                      C.byref(h))
    assert rc == capi.ERR_IROW


@pytest.mark.parametrize("name", ["random_over_damped", "random_over_se", "poisson_48x37_tol"])
def test_one_rank_world_through_comm_init_and_shard_solve(name):
    p, o = CASES[name]
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    eng = EngineSolver(s, 0, p.m, 1, 0)
    d_b = capi.DeviceBuffer.from_array(p.b)
    r = eng.solve(d_b.ptr.value, **o)
    x = eng.d_x.to_array(np.float64, p.n)
    g = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
    assert r.istop == g.istop
    tol = 1e-10 if name != "poisson_48x37_tol" else 1e-5          # (1234 iterations: DESIGN.md 3.3)
    assert np.linalg.norm(x - g.x) <= tol * np.linalg.norm(g.x)
    if name != "poisson_48x37_tol":
        assert r.itn == g.itn and abs(r.anorm - g.anorm) <= 1e-10 * g.anorm
        assert abs(r.rnorm - g.rnorm) <= 1e-10 * g.rnorm
    if o["wantse"]:
        se = eng.d_se.to_array(np.float64, p.n)
        assert np.linalg.norm(se - g.se) <= 1e-9 * np.linalg.norm(g.se)
    r2 = eng.solve(d_b.ptr.value, **o)                              # repeats itself exactly
    assert (r2.itn, r2.anorm, r2.rnorm) == (r.itn, r.anorm, r.rnorm)
    assert np.array_equal(eng.d_x.to_array(np.float64, p.n), x)
    # the handle still solves on its own afterwards
    s.atol, s.btol, s.conlim, s.itnlim = o["atol"], o["btol"], o["conlim"], o["itnlim"]
    r3 = s.solve(p.b, o["damp"])
    assert r3.istop == g.istop


# ---- real RCCL, two ranks on two GPUs (skipped on the one-GPU test boxes; the same calls run there with ranks that
#      SHARE the GPU: test_rccl_ranks_sharing_one_gpu_through_bench_launcher below) ----
def _two_gpus():
    import torch
    return torch.cuda.device_count() >= 2


@pytest.mark.parametrize("overlap", ["0", "1"])
@pytest.mark.parametrize("name", ["random_over_se", "poisson_20x20_it50", "empty_rows_cols_it20"])
def test_rccl_two_gpus_single_process(name, overlap, monkeypatch):
    """(overlap = 1: the exchanges in parts on a second communicator beside the products, shard_engine.h)"""
    if not _two_gpus():
        pytest.skip("needs two GPUs (RCCL refuses ranks that share a device)")
    monkeypatch.setenv("LSQRHIP_SHARD_OVERLAP", overlap)
    p, o = CASES[name]
    h = sharded_handle(p, 2)
    try:
        x, se, istop, itn, sc = solve_handle(h, p, o)
        g = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
        assert (istop, itn) == (g.istop, g.itn)
        assert np.linalg.norm(x - g.x) <= 1e-10 * np.linalg.norm(g.x)
        if o["wantse"]:
            assert np.linalg.norm(se - g.se) <= 1e-9 * np.linalg.norm(g.se)
    finally:
        check(lib().lsqrhip_destroy(h))


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_rccl_two_ranks_through_bench_launcher(overlap):
    """`bench.py --gpus 2` starts its own two ranks (one process per GPU, RCCL), the C++ engine drives them --
    with the exchanges overlapped (LSQRHIP_SHARD_OVERLAP=1: a second communicator split off the first) as well."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "LSQR_BENCH_STRONG_REF": "0", "LSQRHIP_SHARD_OVERLAP": overlap}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
                        "--workload", "random:200000:100000:20"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["steps"] == 20
    assert line["result"]["itn"] == 20 and line["value"] > 0 and line["overlap"] == int(overlap)


@pytest.mark.parametrize("world,overlap,graph", [(2, "0", ""), (2, "1", ""), (3, "0", ""), (4, "1", ""), (6, "0", ""),
                                                 (8, "1", ""), (2, "0", "1"), (3, "0", "1"), (2, "", ""), (3, "", "")])
def test_rccl_ranks_sharing_one_gpu_through_bench_launcher(world, overlap, graph):
    """The RCCL branch at world > 1 on a ONE-GPU box: `bench.py --gpus N` with LSQR_RANKS_SHARE_GPU=1 starts N processes
    that all use device 0; each claims a host of its own (NCCL_HOSTID) so that RCCL takes them, over its socket
    transport on `lo`.  What runs is the real thing above the transport: ncclCommInitRank from an id handed round by
    torch.distributed, the second communicator split off it (overlap = 1), the grouped ncclSend / ncclRecv reduce-scatter,
    both all-gathers, the exchange stream and its events -- driven by the C++ engine, which dist_bench first holds
    against the stage-by-stage Python driver (torch.distributed collectives) on four iterations: `engine` = "c++" with
    no `engine_note` says they agreed on every rank.  The result is then held against ONE handle solving the whole
    matrix.  graph = "1": LSQRHIP_SHARD_GRAPH=1, the batches of 16 iterations captured WITH their RCCL calls and replayed
    (40 iterations: two replays and an eager tail).
    overlap = "" (round 5): nothing pinned in the environment -- the launcher's own form.  The line then carries all three
    schedules (plain, graph, overlap, copy), each probed against the plain engine on four iterations and timed in the same
    invocation, `value` = the best validated one, and `inprocess_sharded_check` (skipped here: one device)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (worlds of 3 and more: 8 iterations -- every exchange of every schedule still runs; over sockets between processes
    #  that share the GPU an iteration costs tens of milliseconds)
    spec, K = "random:200000:100000:20", 40 if graph else (8 if world >= 3 else 20)
    env = {**os.environ, "LSQR_BENCH_STRONG_REF": "0", "LSQRHIP_SHARD_OVERLAP": overlap, "LSQR_RANKS_SHARE_GPU": "1",
           "LSQR_DIST_PROBE_TIMEOUT": "300"}
    if overlap == "":
        env.pop("LSQRHIP_SHARD_OVERLAP")
    else:
        env["LSQR_BENCH_VARIANTS"] = "0"     # (the pinned cases time their one schedule; the "" cases hold the variants)
    if graph:
        env["LSQRHIP_SHARD_GRAPH"] = graph
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", str(K), "--warmup", "2",
           "--workload", spec, "--traffic", "off", "--cpu-iters", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["config"]["world_size"] == world and line["steps"] == K
    assert line["config"]["ranks_share_one_gpu"] is True and "TEST" in line["config"]["backend"]
    assert line["config"]["engine"] == "c++" and line["config"]["engine_note"] is None, line["config"]
    assert line["result"]["itn"] == K and line["value"] > 0
    if overlap == "":     # nothing pinned: every schedule in the one invocation
        v = line["variants"]
        assert set(v) == {"plain", "graph", "overlap", "copy", "overlap_copy"}, v
        for name, e in v.items():
            assert e["validated"] is True and e["value"] > 0, (name, e)
        best = line["config"]["schedule"]
        assert best in v and line["value"] == v[best]["value"] == max(e["value"] for e in v.values())
        assert line["overlap"] == (1 if best in ("overlap", "overlap_copy") else 0) and v["overlap"]["parts"] >= 2
        assert "skipped" in line["inprocess_sharded_check"]        # (one device: the form needs two)
    else:
        assert line["overlap"] == int(overlap) and set(line["variants"]) == {"plain"}
        assert line["config"]["schedule"].startswith("as the environment says") or (overlap == "0" and not graph)
    # the same 20 iterations on one handle that holds the whole matrix (same generator)
    from lsqr_amd import devgen
    from lsqr_amd.capi import DeviceBuffer
    dp = devgen.generate(spec)
    s = dp.solver
    s.atol = s.btol = s.conlim = 0.0
    s.itnlim = K
    d_x = DeviceBuffer(8 * dp.n)
    r = s.solve_device(dp.d_b.ptr.value, d_x.ptr.value, dp.damp)
    tol = 1e-10
    assert r.itn == K
    assert abs(line["result"]["rnorm"] - r.rnorm) <= tol * r.rnorm and abs(line["result"]["anorm"] - r.anorm) <= tol * r.anorm


@pytest.mark.parametrize("fail", ["0", "1"])
def test_bench_distributed_leg_at_world_one_and_its_fallback(fail):
    """bench.py's N > 1 leg forced at world = 1 (one rank over RCCL): the C++ engine by default, and the
    stage-by-stage Python driver when the engine cannot be used (here: a simulated failure)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "LSQR_BENCH_FORCE_DIST": "1", "LSQR_BENCH_STRONG_REF": "0",
           "LSQR_DIST_TEST_ENGINE_FAILURE": fail}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", "29541" if fail == "0" else "29542", os.path.join(root, "bench.py"),
                        "--gpus", "1", "--steps", "12", "--warmup", "2", "--workload", "random:200000:100000:20"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 12 and line["result"]["itn"] == 12
    assert line["config"]["engine"] == ("c++" if fail == "0" else "python")
    assert (line["config"]["engine_note"] is None) == (fail == "0")
    # the N > 1 line carries everything the N = 1 line does: roofline with live PMC traffic of rank
    # 0's block, and the reference's CPU path on a scaled-down instance of the same generator
    from lsqr_amd import dist_bench
    assert set(dist_bench.LINE_KEYS) <= set(line)
    assert set(dist_bench.ROOFLINE_KEYS) <= set(line["roofline"]) and line["roofline"]["traffic"] is not None
    assert line["roofline"]["traffic"] >= 0.9 * line["roofline"]["layout_bytes_per_launch"]   # (what the layout stores)
    assert len(json.dumps(line)) <= 4096
    assert set(dist_bench.CPU_BASELINE_KEYS) <= set(line["cpu_baseline"]) and line["cpu_baseline"]["value"] > 0
    assert line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["kind"] in ("reference", "port")


@pytest.mark.parametrize("csb", ["0", "1"])
def test_loopback_eight_ranks_on_a_larger_system_agree_with_one_gpu(loopback, csb):
    """400k x 150k random, 12 per row (4.8 M nonzeros), damped: eight ranks on one device (loopback exchanges)
    against the ordinary single-handle solve of the same triplets -- same istop and itn, x to 1e-10 -- with
    the ranks' blocks in row windows and, forced, in column-swept row blocks (column splits, exact sums)."""
    import os
    p = P.random_rows(400_000, 150_000, 12, damp=1e-2, seed=5)
    o = dict(damp=p.damp, atol=1e-9, btol=1e-9, conlim=0.0, itnlim=60, wantse=True)
    s1 = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=o["atol"], btol=o["btol"], itnlim=o["itnlim"])
    ref = s1.solve(p.b, p.damp, wantse=True)
    old = os.environ.get("LSQRHIP_CSB")
    os.environ["LSQRHIP_CSB"] = csb
    try:
        h = sharded_handle(p, 8)
    finally:
        os.environ.pop("LSQRHIP_CSB", None)
        if old is not None:
            os.environ["LSQRHIP_CSB"] = old
    try:
        x, se, istop, itn, sc = solve_handle(h, p, o)
        assert (istop, itn) == (ref.istop, ref.itn)
        assert np.linalg.norm(x - ref.x) <= 1e-10 * np.linalg.norm(ref.x)
        assert np.linalg.norm(se - ref.se) <= 1e-9 * np.linalg.norm(ref.se)
        assert abs(sc[0] - ref.anorm) <= 1e-10 * ref.anorm and abs(sc[2] - ref.rnorm) <= 1e-10 * ref.rnorm
    finally:
        check(lib().lsqrhip_destroy(h))


@pytest.mark.parametrize("name", ["poisson_48x37_it100", "empty_rows_cols_it20", "illcond_conlim_it10"])
@pytest.mark.parametrize("ngpu", [1, 3])
def test_sharded_solve_keeps_the_iteration_log(loopback, name, ngpu):
    """`nout /= 0` on a sharded handle (reference src/lsqr.f90:589-595, 813-837, 872-880): the scalars of the
    iteration are replicated and x(1) is the first entry of rank 0's column slice, so rank 0's records ARE the
    reference's log.  On the truncated twins (the reference's own drift < 1e-11 there) the text formatted from a
    3-rank solve must match the reference's own log of the run line for line to the printed digits, and the
    records of the one-GPU path to rounding."""
    import os
    from types import SimpleNamespace

    from cases import assert_log_lines_match
    from lsqr_amd.logfmt import format_log
    p, o = CASES[name]
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"log_{name}.txt")
    h = sharded_handle(p, ngpu)
    try:
        x, se = np.zeros(max(p.n, 1)), np.zeros(max(p.n, 1))
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        b = np.ascontiguousarray(p.b, np.float64)
        check(lib().lsqrhip_solve(h, b.ctypes.data, o["damp"], o["atol"], o["btol"], o["conlim"], o["itnlim"], 0, 1,
                                  x.ctypes.data, None, C.addressof(istop), C.addressof(itn),
                                  *[C.addressof(s) for s in sc]))
        k = lib().lsqrhip_log_count(h)
        rec = np.zeros((k, capi.LOG_STRIDE))
        check(lib().lsqrhip_log_fetch(h, 0, k, rec.ctypes.data))
        ex = np.zeros(6)
        check(lib().lsqrhip_log_extras(h, ex.ctypes.data))
    finally:
        check(lib().lsqrhip_destroy(h))
    assert k > 0 and int(rec[-1][0]) == itn.value
    res = SimpleNamespace(istop=istop.value, itn=itn.value, anorm=sc[0].value, acond=sc[1].value, rnorm=sc[2].value,
                          arnorm=sc[3].value, xnorm=sc[4].value)
    text = format_log(p.m, p.n, o["damp"], False, o["atol"], o["btol"], o["conlim"], o["itnlim"], rec, res,
                      bnorm=ex[0], dxmax=ex[1], maxdx=int(ex[2]), test2_0=ex[5], beta0=ex[4])
    assert_log_lines_match(text.splitlines(), open(gold).read().splitlines(), min(itn.value, 10))
    # ... and against the records of the one-GPU path (another order of the partial sums: rounding apart)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, atol=o["atol"], btol=o["btol"], conlim=o["conlim"],
                                    itnlim=o["itnlim"], nout=os.devnull)
    s.solve(p.b, o["damp"])
    one = s.log_records()
    assert one.shape == rec.shape and np.array_equal(one[:, 0], rec[:, 0]) and np.array_equal(one[:, 11], rec[:, 11])
    assert np.allclose(rec[:, 1:7], one[:, 1:7], rtol=1e-8, atol=0)
    # a solve without the log leaves none behind
    h = sharded_handle(p, ngpu)
    try:
        solve_handle(h, p, dict(o, wantse=False))
        assert lib().lsqrhip_log_count(h) == 0
    finally:
        check(lib().lsqrhip_destroy(h))


@pytest.mark.parametrize("ngpu", [1, 3])
def test_captured_batches_and_eager_launches_agree_bitwise(loopback, ngpu):
    """With one local rank the engine runs its iterations as one hipGraph per batch of 16 (stages and local
    exchanges captured from the rank's stream); LSQRHIP_SHARD_GRAPH=0 enqueues the same launches one by one, as
    groups of several ranks in one process always do.  Same kernels on the same inputs in the same order:
    identical bits, for a run that stops inside a batch, one that ends on a batch boundary and one shorter than
    a batch -- and a captured batch is reused by the next solve."""
    import os
    p, o = CASES["poisson_20x20_it50"]
    for itnlim in (50, 32, 5):
        oo = dict(o, itnlim=itnlim)
        res = {}
        for mode in ("0", None):
            old = os.environ.pop("LSQRHIP_SHARD_GRAPH", None)
            if mode is not None:
                os.environ["LSQRHIP_SHARD_GRAPH"] = mode
            try:
                h = sharded_handle(p, ngpu)
                try:
                    a = solve_handle(h, p, oo)
                    b = solve_handle(h, p, oo)
                finally:
                    check(lib().lsqrhip_destroy(h))
            finally:
                os.environ.pop("LSQRHIP_SHARD_GRAPH", None)
                if old is not None:
                    os.environ["LSQRHIP_SHARD_GRAPH"] = old
            assert np.array_equal(a[0], b[0]) and a[2:] == b[2:]
            res[mode] = a
        assert np.array_equal(res["0"][0], res[None][0]) and res["0"][2:] == res[None][2:]
        assert res[None][3] == itnlim


def test_entry_points_that_need_one_device_refuse_a_sharded_handle(loopback):
    """The parent handle of a sharded system owns no work vectors of its own: the BLAS-1, acheck / xcheck, kernel
    timing and stage entry points must say so (ERR_ARG) instead of launching kernels on null pointers; the
    ordinary solve / aprod / info / log entry points work on it."""
    p, o = CASES["random_over_damped"]
    h = sharded_handle(p, 2)
    try:
        L = lib()
        d = capi.DeviceBuffer(8 * 16)
        res = C.c_double()
        assert L.lsqrhip_dnrm2(h, 16, d.ptr, C.byref(res)) == capi.ERR_ARG
        assert L.lsqrhip_ddot(h, 16, d.ptr, d.ptr, C.byref(res)) == capi.ERR_ARG
        assert L.lsqrhip_dscal(h, 16, C.c_double(2.0), d.ptr) == capi.ERR_ARG
        assert L.lsqrhip_dcopy(h, 16, d.ptr, d.ptr) == capi.ERR_ARG
        inform = C.c_int()
        assert L.lsqrhip_acheck(h, C.c_double(1e-16), C.byref(inform), C.byref(res)) == capi.ERR_ARG
        ms = C.c_double()
        assert L.lsqrhip_bench_kernel(h, 1, 1, C.byref(ms)) == capi.ERR_ARG
        assert L.lsqrhip_shard_begin(h, d.ptr, p.m, 1, 0, C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0), 5, 0,
                                     d.ptr, d.ptr, d.ptr, d.ptr) == capi.ERR_ARG
        # ... and joining a world twice is refused too (the first group would leak)
        x = solve_handle(h, p, o)[0]
        assert np.all(np.isfinite(x))
    finally:
        check(lib().lsqrhip_destroy(h))
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=5)
    check(lib().lsqrhip_shard_comm_init(s._h, 1, 0, 0, p.m, None))
    assert lib().lsqrhip_shard_comm_init(s._h, 1, 0, 0, p.m, None) == capi.ERR_ARG


def test_sharded_blocks_start_at_the_selected_device_and_must_fit_the_node():
    """Row block k of a sharded handle goes to device (selected + k): asking for more blocks than there are
    devices FROM the selected one on fails before any work is done (no loopback here: real devices)."""
    import torch
    have = torch.cuda.device_count()
    p, o = CASES["random_over_damped"]
    irow = np.ascontiguousarray(p.irow, np.int32)
    icol = np.ascontiguousarray(p.icol, np.int32)
    a = np.ascontiguousarray(p.a, np.float64)
    h = C.c_void_p()
    check(lib().lsqrhip_set_device(have - 1))
    try:
        rc = lib().lsqrhip_create_sharded(p.m, p.n, a.size, irow.ctypes.data, icol.ctypes.data, a.ctypes.data, 2, C.byref(h))
        assert rc == capi.ERR_NO_DEVICE and not h.value
        assert b"selected device" in lib().lsqrhip_last_error() or have < 2
    finally:
        check(lib().lsqrhip_set_device(0))


def _solve_with_log(h, p, o):
    x, se = np.zeros(max(p.n, 1)), np.zeros(max(p.n, 1))
    istop, itn = C.c_int(), C.c_int()
    sc = [C.c_double() for _ in range(5)]
    b = np.ascontiguousarray(p.b, np.float64)
    check(lib().lsqrhip_solve(h, b.ctypes.data, o["damp"], o["atol"], o["btol"], o["conlim"], o["itnlim"],
                              int(o["wantse"]), 1, x.ctypes.data, se.ctypes.data if o["wantse"] else None,
                              C.addressof(istop), C.addressof(itn), *[C.addressof(s) for s in sc]))
    k = lib().lsqrhip_log_count(h)
    rec = np.zeros((k, capi.LOG_STRIDE))
    check(lib().lsqrhip_log_fetch(h, 0, k, rec.ctypes.data))
    return x[:p.n].copy(), se[:p.n].copy(), istop.value, itn.value, [s.value for s in sc], rec


@pytest.mark.parametrize("ngpu", [2, 3, 8])
@pytest.mark.parametrize("csb,parts,vmax", [(None, 2, None), ("1", 2, None), ("1", 3, None), ("1", 2, "0"), ("1", 3, "0")])
def test_overlapped_exchanges_change_no_bit(loopback, ngpu, csb, parts, vmax):
    """LSQRHIP_SHARD_OVERLAP=1: the reduce-scatter of T leaves in parts behind the phases of mode 2 and the all-gather
    of v arrives in parts ahead of the phases of mode 1, on an exchange stream of its own (shard_engine.h
    enqueue_iteration_overlap; here in the loopback harness -- the exchanges are device copies, the streams, events
    and parts are the production schedule).  The kernels, their data and the order of every sum are those of the plain
    schedule: x, se, every scalar and every line of the iteration log must be identical to the last bit -- with the
    ranks' blocks in column-swept row blocks built for the parts (stripes, part-major row blocks: the products run
    phase by phase) and in whatever the build chooses at this size (the products run whole, the exchanges in parts).
    vmax = "0" (LSQRHIP_SHARD_VMAX=0: the piece maxima of v do NOT ride with the norms -- also what a world of 69..128 ranks
    gets): mode 1 must then run whole behind all the parts of v (its grids come from a pass over all of V, which phase 0
    would have made while the other parts were still arriving -- round-4 advisor), and still change no bit."""
    import os
    keys = ("LSQRHIP_CSB", "LSQRHIP_SHARD_OVERLAP", "LSQRHIP_SHARD_PARTS", "LSQRHIP_SHARD_VMAX")
    old = {k: os.environ.get(k) for k in keys}
    p = P.random_rows(60000, 12011, 10, seed=31, damp=1e-3)      # (12011: ragged slices and parts)
    o = dict(damp=p.damp, atol=1e-10, btol=1e-10, conlim=0.0, itnlim=25, wantse=True)
    out = []
    try:
        if csb:
            os.environ["LSQRHIP_CSB"] = csb
        os.environ["LSQRHIP_SHARD_PARTS"] = str(parts)
        if vmax is not None:
            os.environ["LSQRHIP_SHARD_VMAX"] = vmax
        for overlap in ("0", "1"):
            os.environ["LSQRHIP_SHARD_OVERLAP"] = overlap
            h = sharded_handle(p, ngpu)
            try:
                first = _solve_with_log(h, p, o)
                again = _solve_with_log(h, p, o)          # ... and repeats itself (nothing left on the exchange streams)
                assert np.array_equal(first[0], again[0]) and first[2:5] == again[2:5]
                out.append(first)
            finally:
                check(lib().lsqrhip_destroy(h))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    off, on = out
    assert (off[2], off[3]) == (on[2], on[3]) and off[3] > 5
    assert np.array_equal(off[0], on[0]) and np.array_equal(off[1], on[1])
    assert off[4] == on[4]
    assert off[5].shape == on[5].shape and np.array_equal(off[5], on[5])
    g = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=p.damp, atol=1e-10, btol=1e-10, itnlim=25)
    assert (on[2], on[3]) == (g.istop, g.itn) and np.linalg.norm(on[0] - g.x) <= 1e-10 * np.linalg.norm(g.x)


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_configs3_as_stated_eight_rccl_ranks_sharing_one_gpu(overlap):
    """BASELINE configs[3] AS STATED -- 10M x 10M at 100 per row (1e9 nonzeros), row blocks on EIGHT ranks, RCCL -- on a
    one-GPU box: the eight processes share device 0 (LSQR_RANKS_SHARE_GPU=1, see the test above).  The C++ engine's
    20 iterations must match the stage-by-stage Python driver's (dist_bench's cross-check: engine "c++", no note) and
    ONE handle that holds the whole matrix, run by the same job on rank 0: anorm and rnorm to 1e-10 (measured:
    2e-16 / 2e-15, profiles/r04/rccl_shared_gpu_configs3_*.json).  Needs ~60 GB of HBM and about a minute."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "LSQR_BENCH_STRONG_REF": "1", "LSQRHIP_SHARD_OVERLAP": overlap, "LSQR_RANKS_SHARE_GPU": "1",
           "LSQR_DIST_PROBE_TIMEOUT": "600", "LSQR_BENCH_VARIANTS": "0"}
    # (6 iterations prove the path -- 8 ranks, every exchange, the one-handle comparison; the socket transport between
    #  processes that share the GPU is what takes the time: round 5 ran 20)
    K = 6
    detail = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bench_detail_w8_{overlap}.json")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", str(K), "--warmup", "1",
                        "--workload", "random:10000000:10000000:100", "--traffic", "off", "--cpu-iters", "0",
                        "--detail", detail],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    raw = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    assert len(raw) <= 4096
    line = json.loads(raw)
    assert line["n_gpus"] == 8 and line["config"]["ranks_share_one_gpu"] is True and line["overlap"] == int(overlap)
    assert line["config"]["engine"] == "c++" and line["config"]["engine_note"] is None, line["config"]
    assert line["config"]["rows_per_rank"] == [1250000] * 8 and line["result"]["itn"] == K
    assert line["config"]["workload"].startswith("random:10000000:10000000:100 m=10000000 n=10000000 nnz=1000000000 damp=0.001")
    with open(detail) as f:
        ref = json.load(f)["strong_scaling_ref"]
    assert ref["result"]["itn"] == K and ref["result"]["istop"] == line["result"]["istop"]
    assert ref["sharded_vs_1gpu"] == line["sharded_vs_1gpu"] and line["value_1gpu_same_workload"] == ref["value"]
    assert ref["sharded_vs_1gpu"]["rnorm_rel"] <= 1e-10 and ref["sharded_vs_1gpu"]["anorm_rel"] <= 1e-10


def test_python_stage_driver_over_nccl_ranks_sharing_one_gpu():
    """The fall-back of bench.py's distributed leg at world > 1: the C++ engine "fails" on every rank (simulated), the
    ranks agree on it and the stage-by-stage Python driver runs the timed solve -- lsqr_amd.dist.ShardedLSQR with its
    scalar all-reduce, slice scatter and gather as torch.distributed collectives on the nccl backend (three ranks that
    share device 0).  Same 20 iterations as one handle on the whole matrix."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "LSQR_BENCH_STRONG_REF": "1", "LSQR_RANKS_SHARE_GPU": "1", "LSQR_DIST_TEST_ENGINE_FAILURE": "1"}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "20", "--warmup", "2",
                        "--workload", "random:200000:100000:20", "--traffic", "off", "--cpu-iters", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["config"]["engine"] == "python" and "LSQR_DIST_TEST_ENGINE_FAILURE" in line["config"]["engine_note"]
    assert line["result"]["itn"] == 20
    d = line["sharded_vs_1gpu"]
    assert d["rnorm_rel"] <= 1e-10 and d["anorm_rel"] <= 1e-10


@pytest.mark.parametrize("ngpu,overlap", [(3, "0"), (8, "0"), (5, "1")])
def test_exchanges_as_copies_on_a_stream_per_peer_change_no_bit(ngpu, overlap):
    """The copy form of the exchanges (what LSQRHIP_SHARD_COPY=1 offers a one-process group on a node -- peer copies,
    no CU set aside -- and what the loopback harness runs on one device): every rank pulls from each peer on a copy
    stream of that peer's, tied to the rank's stream by events (shard_engine.h PullBatch), so that pulls from different
    peers are independent commands.  With LSQRHIP_SHARD_COPY_STREAMS=0 the pulls queue on the rank's own stream, one
    after the other: x, se, the scalars and the log must not differ by a bit.  (LSQRHIP_SHARD_COPY=1 itself needs one
    device per rank; together with the harness switch it takes the harness's device mapping.)"""
    import os
    keys = ("LSQRHIP_SHARD_LOOPBACK", "LSQRHIP_SHARD_COPY", "LSQRHIP_SHARD_COPY_STREAMS", "LSQRHIP_SHARD_OVERLAP")
    old = {k: os.environ.get(k) for k in keys}
    p = P.random_rows(30000, 7001, 9, seed=37, damp=1e-3)
    o = dict(damp=p.damp, atol=1e-10, btol=1e-10, conlim=0.0, itnlim=20, wantse=True)
    out = []
    try:
        os.environ["LSQRHIP_SHARD_LOOPBACK"] = "1"
        os.environ["LSQRHIP_SHARD_COPY"] = "1"
        os.environ["LSQRHIP_SHARD_OVERLAP"] = overlap
        for streams in ("1", "0"):
            os.environ["LSQRHIP_SHARD_COPY_STREAMS"] = streams
            h = sharded_handle(p, ngpu)
            try:
                out.append(_solve_with_log(h, p, o))
            finally:
                check(lib().lsqrhip_destroy(h))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    a, b = out
    assert (a[2], a[3]) == (b[2], b[3]) and a[3] > 5 and a[4] == b[4]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[5], b[5])
    g = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=p.damp, atol=1e-10, btol=1e-10, itnlim=20)
    assert (a[2], a[3]) == (g.istop, g.itn) and np.linalg.norm(a[0] - g.x) <= 1e-10 * np.linalg.norm(g.x)


def test_copy_mode_without_a_device_per_rank_fails_loudly():
    """LSQRHIP_SHARD_COPY=1 is for ranks on devices of their own: three ranks on a one-GPU box (and no harness switch)
    is the no-device error, as without it."""
    import os
    import torch
    if torch.cuda.device_count() >= 3:
        pytest.skip("this node has a device per rank")
    old = {k: os.environ.get(k) for k in ("LSQRHIP_SHARD_COPY", "LSQRHIP_SHARD_LOOPBACK")}
    os.environ["LSQRHIP_SHARD_COPY"] = "1"
    os.environ.pop("LSQRHIP_SHARD_LOOPBACK", None)
    try:
        p, _ = CASES["random_over_damped"]
        with pytest.raises(capi.LsqrHipError) as ei:
            sharded_handle(p, 3)
        assert ei.value.code == capi.ERR_NO_DEVICE
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


# ---- an engine solve that fails half way must not steer the next caller of the C stages (round-4 advisor) ----
class _OneRankComm:
    """lsqr_amd.dist.TorchComm's interface for a world of one, without a process group."""
    world, rank, backend = 1, 0, "none"

    def all_reduce_scalars(self, t):
        return

    def scatter_slices(self, T, R, chunk):
        R[:chunk].copy_(T[:chunk])

    def gather_slices(self, V, chunk):
        return

    def agree_max(self, value):
        return int(value)

    def barrier(self):
        return


def _flags(s):
    v = C.c_int64()
    check(lib().lsqrhip_get_option(s._h, b"shard_engine_flags", C.byref(v)))
    return int(v.value)


@pytest.mark.parametrize("csb", [None, "1"])
def test_stage_driver_after_an_engine_solve_that_failed_half_way(monkeypatch, csb):
    """The engine sets `own slice stays in T`, `norms are gathered by me`, `sums is the long message` on the rank's handle
    and lowers them in lsqrhip_shard_end.  An error exit in between (here: LSQRHIP_SHARD_FAIL_AT, the test hook; in the
    field: a failing RCCL call, the did-not-terminate guard) used to leave them up, and the Python stage driver that
    lsqr_amd/dist_bench.py falls back to -- same handle, a 4-double `sums` -- then read stale gathered norms and had
    512 bytes cleared in its 32-byte tensor."""
    import torch
    from lsqr_amd.dist import HipShardBackend, ShardedLSQR
    p, o = CASES["random_over_se"]
    if csb:
        monkeypatch.setenv("LSQRHIP_CSB", csb)

    def fresh():
        return lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)

    def stage_driver(s):
        be = HipShardBackend(s, p.m, 1, 0)
        try:
            d_b = torch.tensor(p.b, dtype=torch.float64, device="cuda")
            r = ShardedLSQR(be, _OneRankComm(), poll_every=3).solve(d_b.data_ptr(), **o)
            torch.cuda.synchronize()
            return r.istop, r.itn, r.anorm, r.rnorm, r.x.cpu().numpy().copy(), r.se.cpu().numpy().copy()
        finally:
            be.close()

    clean = stage_driver(fresh())                      # a handle the engine never touched
    s = fresh()
    eng = EngineSolver(s, 0, p.m, 1, 0)
    d_b = capi.DeviceBuffer.from_array(p.b)
    monkeypatch.setenv("LSQRHIP_SHARD_FAIL_AT", "8")
    with pytest.raises(capi.LsqrHipError, match="injected failure"):
        eng.solve(d_b.ptr.value, **o)
    monkeypatch.delenv("LSQRHIP_SHARD_FAIL_AT")
    assert _flags(s) == 0                              # lowered by the engine's own error exit
    after = stage_driver(s)
    assert after[:4] == clean[:4]
    assert np.array_equal(after[4], clean[4]) and np.array_equal(after[5], clean[5])
    g = oracle.port().solve(p.m, p.n, p.irow, p.icol, p.a, p.b, **o)
    assert after[0] == g.istop and after[1] == g.itn
    assert np.linalg.norm(after[4] - g.x) <= 1e-10 * np.linalg.norm(g.x)
    assert abs(after[2] - g.anorm) <= 1e-10 * g.anorm and abs(after[3] - g.rnorm) <= 1e-10 * g.rnorm
    # ... and the engine still works on the handle afterwards, and leaves nothing behind when it succeeds
    r = eng.solve(d_b.ptr.value, **o)
    assert (r.istop, r.itn) == (g.istop, g.itn) and _flags(s) == 0
    # belt and braces: flags left up by hand (as a crash between begin and end would) are cleared by the next begin
    # of a caller that is not the engine
    be = HipShardBackend(s, p.m, 1, 0)
    try:
        dbt = torch.tensor(p.b, dtype=torch.float64, device="cuda")
        be.run(lambda: be.begin(dbt.data_ptr(), o["damp"], o["atol"], o["btol"], o["conlim"], o["itnlim"], o["wantse"]))
        assert _flags(s) == 8                          # open, nothing of the engine's
        be.run(lambda: be.end(_OneRankComm()))
        assert _flags(s) == 0
    finally:
        be.close()


@pytest.mark.parametrize("ngpu", [1, 3])
@pytest.mark.parametrize("name", ["random_over_se", "poisson_20x20_it50", "empty_rows_cols_it20", "t1_readme_damped"])
def test_step_two_inside_the_update_launch_changes_no_bit(loopback, name, ngpu, monkeypatch):
    """Round 5: the engine's scalar step 2 (alpha, the rotations, t1..t3) runs inside the slice update's launch -- every
    workgroup evaluates the rotation itself from inputs the launch does not write, workgroup 0 is the scalar machine
    (shard_api.h k_update_slice_g) -- instead of a one-workgroup kernel in front of it (LSQRHIP_SHARD_FUSE_S2=0).  Same
    functions on the same inputs: x, se, every scalar and every record of the iteration log must be identical."""
    p, o = CASES[name]
    out = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("LSQRHIP_SHARD_FUSE_S2", fuse)
        h = sharded_handle(p, ngpu)
        try:
            out.append(_solve_with_log(h, p, o))
        finally:
            check(lib().lsqrhip_destroy(h))
    a, b = out
    assert a[2:5] == b[2:5]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[5], b[5])
