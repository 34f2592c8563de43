"""bench.py's N > 1 leg: the row-block sharded solve, one rank per GPU over RCCL.

Default workload: BASELINE.json configs[3] made concrete per SURVEY.md section 8d --
m = n = 10^7 random sparse least squares with 100 nonzeros per row (10^9 nonzeros, density
1e-5; the literal "~1 %" is 10^12 nonzeros = 12 TB and cannot exist on 8 x 288 GB), damp = 1e-3.
Total work is fixed as N grows: "scaling": "strong".  The matrix is generated in HBM, each
rank only its own row block.  `bench.py --gpus 1` measures the SAME system whole on one GPU: the lines of a series
divide directly.  The one JSON line is at most 4 KB (bench.emit); everything else goes to the detail file it names.
"""
from __future__ import annotations

import json
import os
import time

DEFAULT_SPEC = "random:10000000:10000000:100"
CPU_SAMPLE_NNZ = 20_000_000     # the CPU baseline runs a scaled-down instance of the same generator (~10-20 s)

# keys of the one JSON line (tests/test_capi_cpu.py holds both line shapes to their key sets)
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "result", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample")


def cpu_sample_spec(cfg: dict) -> str:
    """The same generator at a size the reference's single-threaded CPU path finishes in seconds: rows and
    columns divided by the same factor, nonzeros per row kept (per-iteration cost is linear in nnz)."""
    if cfg["kind"] != "random":
        return ""
    f = max(1, int(round(cfg["m"] * cfg["per_row"] / CPU_SAMPLE_NNZ)))
    return f"random:{max(cfg['m'] // f, 1)}:{max(cfg['n'] // f, 1)}:{cfg['per_row']}"


def cpu_baseline_scaled(spec: str, nnz_full: int, iters: int = 20):
    """`cpu_baseline` of the N > 1 line: oracle/_ref (or the C port) on a scaled-down instance of the SAME
    generator, one core; `value` is the measured rate scaled linearly in nnz to the full workload (an
    iteration of LSQR is O(nnz)); the raw sample is quoted beside it."""
    import time as _t

    import oracle

    from . import devgen
    cfg = devgen.parse_spec(spec)
    sspec = cpu_sample_spec(cfg)
    if not sspec:
        return None
    irow, icol, a, b = devgen.download_coo(sspec)      # generated in HBM (bit-identical to lsqr_amd.problems), copied out
    sc = devgen.parse_spec(sspec)
    rf = oracle.ref()
    eng, kind = (rf, "reference") if rf is not None else (oracle.port(), "port")
    t0 = _t.perf_counter()
    r = eng.solve(sc["m"], sc["n"], irow, icol, a, b, damp=sc["damp"], itnlim=iters)
    dt = _t.perf_counter() - t0
    rate = r.itn / dt
    return {"value": rate * len(a) / nnz_full, "unit": "it/s", "cores": 1, "kind": kind,
            "value_is": "the sample's measured rate scaled by nnz(sample) / nnz(workload): LSQR's iteration is O(nnz)",
            "sample_value": rate, "sample_nnz": int(len(a)),
            "sample": f"{sspec} (same generator and seed, rows and columns scaled down, {len(a)} nonzeros): {r.itn} "
                      f"iterations in {dt:.2f} s, "
                      f"{'oracle/_ref (reference compiled with amdflang -O2)' if kind == 'reference' else 'oracle C port'}"}


def rank0_block_traffic(spec: str, row0: int, nrows: int):
    """HBM bytes per mode-1 product of rank 0's row block from rocprofv3 PMC passes run as child processes of
    rank 0 (bench.py live_traffic), BEFORE this process joins the world (the other ranks wait in the
    rendezvous meanwhile).  (traffic dict or None, note)"""
    import bench
    plan = [[spec, {}, int(row0), int(nrows)]]
    # (120 s per pass: the other ranks wait under the 600 s default of init_process_group)
    traffic, note = bench.live_traffic(plan, timeout=120)
    return traffic.get((spec, "{}", (int(row0), int(nrows)))), note


def share_one_gpu(rank: int) -> bool:
    """LSQR_RANKS_SHARE_GPU=1 (a TEST switch, never a measurement): every rank of the job uses device 0.  RCCL refuses
    ranks that share a device of one HOST, so each rank claims a host of its own (NCCL_HOSTID) and the ranks talk over the
    loopback interface: the socket transport instead of xGMI, but the same communicator set-up, the same grouped
    send / receive and all-gather calls, the same stream ordering -- what a one-GPU box can execute of the N > 1 path.
    Must run before anything initialises RCCL (torch's or the library's)."""
    if os.environ.get("LSQR_RANKS_SHARE_GPU", "0") in ("", "0"):
        return False
    os.environ["NCCL_HOSTID"] = f"lsqr-shared-gpu-rank{rank}"
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    return True


class _Env:
    """LSQRHIP_* knobs for the duration of a build or a solve."""

    def __init__(self, env):
        self.env = env or {}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def timed_sharded_solve(drv, d_b, K, kw, dist, torch, spec="", env=None):
    """EXACTLY K iterations between barrier + synchronize on both sides; the slowest rank's time.
    (dt, last result, restarts)"""
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # exactly K iterations: configs[3] does not stop on its own in 5000 (scripts/converge_at.py); a
    # --workload that reaches machine precision earlier is started again on the same b (every rank
    # sees the same itn, so the ranks stay in step)
    done, restarts = 0, 0
    with _Env(env):
        while done < K:
            r = drv.solve(d_b, itnlim=K - done, **kw)
            done += r.itn
            if done < K:
                restarts += 1
                if r.itn == 0:
                    raise SystemExit(f"bench.py: workload {spec} stops at iteration 0")
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert done == K and (restarts > 0 or r.istop == 5), (done, r.itn, r.istop)
    return float(t.item()), r, restarts


def all_ranks_ok(ok: bool, dist, torch) -> bool:
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def inprocess_sharded_check(world: int):
    """Rank 0 only, after the timed region: the ONE-process form of the sharded solve -- lsqrhip_create_sharded(ngpu = N),
    what Fortran's `initialize(..., ngpu = N)` binds (lsqr_amd/fortran/lsqr_module.f90; ncclCommInitAll + peer copies or
    RCCL between the devices of this process) -- on a small system, against one handle of the same system.  GPUTEST boxes
    have one GPU: this is the only place that form meets several real devices."""
    import ctypes as C

    import numpy as np
    import torch

    from . import capi
    from . import problems as P
    from .solver import lsqr_solver_ez
    have = torch.cuda.device_count()
    ngpu = min(world, have)
    if ngpu < 2:
        return {"skipped": f"this process sees {have} device(s): the one-process sharded form needs two"}
    p = P.random_rows(400000, 100000, 20, damp=1e-3)
    K = 30
    t0 = time.perf_counter()
    one = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=K)
    r1 = one.solve(p.b, p.damp)
    h = C.c_void_p()
    irow = np.ascontiguousarray(p.irow, np.int32)
    icol = np.ascontiguousarray(p.icol, np.int32)
    a = np.ascontiguousarray(p.a, np.float64)
    capi.check(capi.lib().lsqrhip_create_sharded(p.m, p.n, a.size, irow.ctypes.data, icol.ctypes.data, a.ctypes.data, ngpu,
                                                 C.byref(h)))
    try:
        x = np.zeros(p.n)
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        b = np.ascontiguousarray(p.b, np.float64)
        capi.check(capi.lib().lsqrhip_solve(h, b.ctypes.data, p.damp, 0.0, 0.0, 0.0, K, 0, 0, x.ctypes.data, None,
                                            C.addressof(istop), C.addressof(itn), *[C.addressof(v) for v in sc]))
    finally:
        capi.lib().lsqrhip_destroy(h)
    relx = float(np.linalg.norm(x - r1.x) / np.linalg.norm(r1.x))
    out = {"ngpu": ngpu, "workload": f"{p.name}: {K} iterations", "istop": [int(istop.value), int(r1.istop)],
           "itn": [int(itn.value), int(r1.itn)], "rel_dx": relx,
           "anorm_rel": abs(sc[0].value - r1.anorm) / r1.anorm, "rnorm_rel": abs(sc[2].value - r1.rnorm) / r1.rnorm,
           "seconds": time.perf_counter() - t0}
    out["ok"] = bool(out["istop"][0] == out["istop"][1] and out["itn"][0] == out["itn"][1] and relx <= 1e-10
                     and out["anorm_rel"] <= 1e-10 and out["rnorm_rel"] <= 1e-10)
    return out


def run_distributed(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    from . import capi, devgen
    from .dist import EngineSolver, HipShardBackend, ShardedLSQR, TorchComm, partition_rows

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    shared = share_one_gpu(rank)
    if shared:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no MI355X visible; the HIP path has no CPU fallback")
    spec = DEFAULT_SPEC if args.workload == "auto" else args.workload
    cfg = devgen.parse_spec(spec)
    blocks = partition_rows(cfg["m"], world, devgen.row_weights(cfg))
    traffic, traffic_note = None, "--traffic off"
    if rank == 0 and args.traffic == "live" and not args.no_roofline:
        try:
            traffic, traffic_note = rank0_block_traffic(spec, *blocks[0])
        except Exception as e:  # noqa: BLE001  (never fail the measurement over its annotation)
            traffic, traffic_note = None, f"PMC passes failed: {e!r}"
    torch.cuda.set_device(local)
    capi.check(capi.lib().lsqrhip_set_device(local))
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    comm = TorchComm()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")

    K, W = args.steps, args.warmup
    row0, nrows = blocks[rank]
    # LSQRHIP_SHARD_OVERLAP=1 (on every rank): the rank's layouts are built for exchanges in parts -- the build must
    # know the world it is a block of (csrc/lsqrhip.hip finish_create)
    if os.environ.get("LSQRHIP_SHARD_OVERLAP", "0") not in ("", "0"):
        os.environ["LSQRHIP_SHARD_WORLD"] = str(world)
    prob = devgen.generate(spec, row0, nrows)
    # the loop itself -- kernels and RCCL calls -- runs in C++ (csrc/shard_engine.h); LSQR_DIST_ENGINE=python
    # selects the stage-by-stage driver over torch.distributed instead (same stages, same arithmetic).
    # The C++ engine's RCCL calls have run at world > 1 only between processes that share one GPU (share_one_gpu: socket
    # transport) and through the in-process loopback -- never over xGMI -- so at world > 1 it is first checked against the Python driver
    # on a 4-iteration solve -- the two must agree to rounding -- and any failure or disagreement, on any
    # rank, falls back to the Python driver for the timed run (reported in config.engine).
    engine = os.environ.get("LSQR_DIST_ENGINE", "c++")
    kw = dict(damp=cfg["damp"], atol=0.0, btol=0.0, conlim=0.0)
    be, drv, engine_note = None, None, None

    def python_driver():
        b = HipShardBackend(prob.solver, cfg["m"], world, rank)
        return b, ShardedLSQR(b, comm, poll_every=min(16, max(1, K)))

    if engine != "python":
        ok = 1
        # A rank that hangs inside RCCL (a peer died in ncclCommInitRank, a send without its receive) cannot be
        # recovered in-process: the watchdog ends THIS process with a non-zero code and the launcher tears the job
        # down -- no restart, no exec from a process that holds the GPU.
        import threading
        watchdog = threading.Timer(float(os.environ.get("LSQR_DIST_PROBE_TIMEOUT", "600")), lambda: os._exit(3))
        watchdog.daemon = True
        watchdog.start()
        try:
            if os.environ.get("LSQR_DIST_TEST_ENGINE_FAILURE") == "1":   # (tests: the fall-back path)
                raise RuntimeError("LSQR_DIST_TEST_ENGINE_FAILURE")
            drv = EngineSolver(prob.solver, row0, cfg["m"], world, rank)   # (its handshake is collective-safe: dist.py)
            if world > 1:
                r_eng = drv.solve(prob.d_b.ptr.value, itnlim=4, **kw)
        except Exception as e:  # noqa: BLE001  (RCCL missing / refusing: report and use the other driver)
            engine_note, ok, drv = f"c++ engine failed: {e!r}", 0, None
        finally:
            watchdog.cancel()
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                be, pdrv = python_driver()
                r_py = pdrv.solve(prob.d_b.ptr.value, itnlim=4, **kw)
                same = (r_eng.itn == r_py.itn and abs(r_eng.rnorm - r_py.rnorm) <= 1e-12 * abs(r_py.rnorm)
                        and abs(r_eng.anorm - r_py.anorm) <= 1e-12 * abs(r_py.anorm))
                flag = torch.tensor([1 if same else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    be.close()
                    be = None
                else:
                    engine_note = (f"c++ engine disagreed with the python driver after 4 iterations "
                                   f"(rnorm {r_eng.rnorm!r} vs {r_py.rnorm!r}): python driver used")
                    drv, engine = pdrv, "python"
            else:
                drv = None
    if drv is None:
        be, drv = python_driver()
        engine = "python"

    if W > 0:
        drv.solve(prob.d_b.ptr.value, itnlim=W, **kw)
    dt, r, restarts = timed_sharded_solve(drv, prob.d_b.ptr.value, K, kw, dist, torch, spec)
    # (what each schedule is: DESIGN.md section 5; the line carries value / ms_per_step / validated only)
    variants = {"plain": {"value": K / dt, "ms_per_step": 1e3 * dt / K, "validated": True}}

    nnz_all = torch.tensor([prob.nnz], dtype=torch.int64, device="cuda")
    dist.all_reduce(nnz_all)
    nnz_total = int(nnz_all.item())

    # local SpMV rate of this rank's block (roofline of the dominant kernel): PHYSICAL bytes of the
    # layout in use / average launch time; the SURVEY 8d algorithmic rate beside it, labelled
    if be is not None:
        be.close()
    s = prob.solver
    reps = max(10, min(K, 100))
    avg1 = s.bench_kernel(1, reps)
    info = s.info()
    b1 = 12 * prob.nnz + info["rowptr_bytes"] * (nrows + 1) + 8 * cfg["n"] + 16 * nrows
    lay1 = info["csr_bytes"] + 8 * cfg["n"] + 16 * nrows
    ach = lay1 / (avg1 * 1e-3) / 1e9
    kname = "k_spmv_csb" if info["xlds"] == 3 else ("k_spmv_fused + k_panel_combine" if info["panels"] > 1 else "k_spmv_*")

    # Same workload on ONE GPU (rank 0), outside the timed region: N = 1 of bench.py runs a
    # different configuration (BASELINE configs[1]), so the strong-scaling speedup of THIS
    # problem is made self-contained here.
    ref = None
    sref = os.environ.get("LSQR_BENCH_STRONG_REF", "1")
    if rank == 0 and ((sref == "1" and world > 1) or sref == "force"):
        try:
            import ctypes as C
            full = devgen.generate(spec)
            d_x = capi.DeviceBuffer(8 * max(cfg["n"], 1))
            full.solver.atol = full.solver.btol = full.solver.conlim = 0.0
            kr = max(2, min(K, 50))
            full.solver.itnlim = max(2, min(W, 4))
            full.solver.solve_device(full.d_b.ptr.value, d_x.ptr.value, cfg["damp"])
            full.solver.itnlim = kr
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rr = full.solver.solve_device(full.d_b.ptr.value, d_x.ptr.value, cfg["damp"])
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            ref = {"n_gpus": 1, "steps": kr, "value": rr.itn / dt1, "unit": "it/s", "ms_per_step": 1e3 * dt1 / rr.itn,
                   "note": "same matrix, whole on rank 0's GPU, measured after the timed sharded solve",
                   "result": {"istop": rr.istop, "itn": rr.itn, "anorm": rr.anorm, "rnorm": rr.rnorm}}
            if kr == K and restarts == 0 and rr.itn == r.itn:
                # the same K iterations on one handle and on `world` ranks: how far the run's norms lie apart
                ref["sharded_vs_1gpu"] = {"rnorm_rel": abs(r.rnorm - rr.rnorm) / abs(rr.rnorm) if rr.rnorm else 0.0,
                                          "anorm_rel": abs(r.anorm - rr.anorm) / abs(rr.anorm) if rr.anorm else 0.0}
            del full, d_x
        except Exception as e:  # e.g. not enough HBM for the whole matrix: report, do not fail the run
            ref = {"error": repr(e)}

    cpu = None
    if rank == 0 and args.cpu_iters > 0:
        try:
            cpu = cpu_baseline_scaled(spec, nnz_total)
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}

    # RCCL (NCCL_DEBUG=VERSION on the GPU boxes) writes its banner to STDOUT through C stdio,
    # which sits in a buffer until the process exits -- i.e. after the JSON line.  Drain every
    # rank's C buffers first so that the JSON line is the last thing this job prints on stdout.
    import ctypes
    import sys
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    dist.barrier()

    detail = {}
    if rank == 0:
        import bench as _bench
        b2 = 12 * prob.nnz + info["rowptr_bytes"] * (cfg["n"] + 1) + 8 * nrows + 16 * cfg["n"]
        try:
            avg2 = s.bench_kernel(2, reps)
        except Exception:  # noqa: BLE001
            avg2 = None
        ach_alg = b1 / (avg1 * 1e-3) / 1e9
        out = {
            "metric": "lsqr_iterations_per_sec", "value": K / dt, "unit": "it/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{spec} m={cfg['m']} n={cfg['n']} nnz={nnz_total} damp={cfg['damp']} "
                                   f"({_bench.HEADLINE_NOTE}, row-block sharded over {world} GPU{'s' if world > 1 else ''})",
                       "rows_per_rank": ([b[1] for b in blocks] if world <= 8 else
                                         {"min": min(b[1] for b in blocks), "max": max(b[1] for b in blocks)}),
                       "backend": ("nccl (RCCL; TEST: all ranks on ONE GPU, socket transport over lo -- not a measurement)" if shared
                                   else "nccl (RCCL over xGMI)"), "ranks_share_one_gpu": shared, "engine": engine,
                       "engine_note": engine_note[:200] if engine_note else None, "world_size": dist.get_world_size(),
                       "restarts": restarts},
            "result": {"istop": r.istop, "itn": r.itn, "anorm": r.anorm, "rnorm": r.rnorm},
            # the dominant kernel on rank 0's row block, on SURVEY 8d's algorithmic bytes of that block (as at N = 1)
            "roofline": {"bound": "cache" if ach_alg > 8000.0 else "hbm",
                         "kernel": f"{kname} (aprod mode 1, local row block, rank 0)",
                         "achieved": ach_alg, "peak": 8000.0, "unit": "GB/s", "frac": ach_alg / 8000.0,
                         "traffic": traffic.get("bytes_per_launch") if traffic else None,
                         "bytes_per_launch": b1, "bytes_are": "SURVEY 8d B1 of the block (algorithmic)",
                         "avg_launch_us": avg1 * 1e3, "launches": reps,
                         "frac_layout": ach / 8000.0, "layout_bytes_per_launch": lay1,
                         "frac_mode2": (b2 / (avg2 * 1e-3) / 1e9 / 8000.0) if avg2 else None,
                         "avg_launch_us_mode2": avg2 * 1e3 if avg2 else None},
        }
        if traffic is None:
            out["roofline"]["traffic_note"] = (traffic_note or "")[:160]
        detail["traffic_detail"] = traffic
        detail["exchanges_per_iteration"] = {"allreduce_scalars": "1 + 2 doubles (all-gather + rank-ordered sum)",
                                             "reduce_scatter_bytes_out_per_gpu": 8 * cfg["n"] * (world - 1) // world,
                                             "allgather_bytes_in_per_gpu": 8 * cfg["n"] * (world - 1) // world}
        # next to `value`: the same matrix whole on ONE GPU, measured by this run, and the ratio (bench.py --gpus 1
        # measures the same system: its `value` is this figure from a run of its own)
        out["value_1gpu_same_workload"] = ref["value"] if ref and "value" in ref else None
        out["speedup_vs_1gpu_same_workload"] = (K / dt) / ref["value"] if ref and "value" in ref else None
        # the schedule the engine of the line's `value` REALLY ran (a requested overlap that could not be set up is off)
        out["overlap"] = (int(prob.solver.get_option("shard_overlap")) if engine == "c++" else 0)
        out["cpu_baseline"] = cpu
        if ref is not None:
            detail["strong_scaling_ref"] = ref
            if "sharded_vs_1gpu" in ref:
                out["sharded_vs_1gpu"] = ref["sharded_vs_1gpu"]
        out["variants"] = variants
        out["config"]["schedule"] = "plain"
    else:
        out = None

    # ---- the other schedules of the same solve, in the same invocation (round 5) -------------------------------------
    # graph   : the plain schedule's batches captured in a hipGraph, RCCL's kernels with them (LSQRHIP_SHARD_GRAPH=1)
    # overlap : the two n-vector exchanges in parts on a stream of their own beside the products (LSQRHIP_SHARD_OVERLAP=1:
    #           the rank's layouts are rebuilt for it, a second communicator is split off)
    # copy    : the plain schedule, exchanges as copy-engine pulls over IPC-mapped buffers (LSQRHIP_SHARD_COPY=1)
    # Each is probed first -- 4 iterations against the plain engine's, which was held to the Python driver above: they
    # must agree to 1e-12 on every rank -- then timed like the plain one.  `value` = the best VALIDATED schedule.
    # None of this has run over xGMI before the driver's multi-GPU run: a hang inside RCCL cannot be recovered from, so
    # a watchdog prints the line as it stands (plain schedule, the failing variant named) and ends the job cleanly.
    # (a run with LSQRHIP_SHARD_OVERLAP / LSQRHIP_SHARD_GRAPH set in its environment measures that one schedule)
    pinned = [k for k in ("LSQRHIP_SHARD_OVERLAP", "LSQRHIP_SHARD_GRAPH", "LSQRHIP_SHARD_COPY")
              if os.environ.get(k, "0") not in ("", "0")]
    if rank == 0 and pinned:
        out["config"]["schedule"] = "as the environment says: " + ", ".join(f"{k}={os.environ[k]}" for k in pinned)
    want_variants = engine == "c++" and not pinned and (world > 1 or os.environ.get("LSQR_BENCH_VARIANTS") == "1") and \
        os.environ.get("LSQR_BENCH_VARIANTS", "1") != "0"
    state = {"doing": None}

    def bail():
        if rank == 0 and out is not None:
            out["variants"][state["doing"] or "?"] = {"error": "timed out (watchdog): RCCL hang?", "validated": False}
            out["config"]["variants_note"] = f"stopped by the watchdog inside `{state['doing']}`"
            _bench.emit(out, detail, args.detail)
        os._exit(3)     # a hung job is a FAILED job on every rank: the line above says what was measured before it

    if want_variants:
        import threading
        wd = threading.Timer(float(os.environ.get("LSQR_DIST_VARIANT_TIMEOUT", "420")), bail)
        wd.daemon = True
        wd.start()
        r_ref = drv.solve(prob.d_b.ptr.value, itnlim=4, **kw)     # the plain engine's 4 iterations (same on every rank)

        results = {}

        def agrees(rv):
            return (rv.itn == r_ref.itn and abs(rv.rnorm - r_ref.rnorm) <= 1e-12 * abs(r_ref.rnorm)
                    and abs(rv.anorm - r_ref.anorm) <= 1e-12 * abs(r_ref.anorm))

        def measure(name, vdrv, vprob, env, what):
            state["doing"] = name
            ok, note = True, None
            try:
                with _Env(env):
                    rv = vdrv.solve(vprob.d_b.ptr.value, itnlim=4, **kw)
                ok = agrees(rv)
                if not ok:
                    note = f"disagrees with the plain schedule after 4 iterations (rnorm {rv.rnorm!r} vs {r_ref.rnorm!r})"
            except Exception as e:  # noqa: BLE001
                ok, note = False, repr(e)
            if not all_ranks_ok(ok, dist, torch):
                variants[name] = {"validated": False, "error": (note or "failed or disagreed on another rank")[:120]}
                return
            with _Env(env):
                if W > 0:
                    vdrv.solve(vprob.d_b.ptr.value, itnlim=W, **kw)
            dtv, rv, _ = timed_sharded_solve(vdrv, vprob.d_b.ptr.value, K, kw, dist, torch, spec, env)
            variants[name] = {"value": K / dtv, "ms_per_step": 1e3 * dtv / K, "validated": True}
            results[name] = {"istop": rv.istop, "itn": rv.itn, "anorm": rv.anorm, "rnorm": rv.rnorm}

        measure("graph", drv, prob, {"LSQRHIP_SHARD_GRAPH": "1"},
                "the plain schedule, batches of iterations captured in a hipGraph (RCCL's kernels with them)")
        prob_ov, drv_ov, ok = None, None, True
        state["doing"] = "overlap (build + communicator)"
        try:
            with _Env({"LSQRHIP_SHARD_OVERLAP": "1", "LSQRHIP_SHARD_WORLD": str(world)}):
                prob_ov = devgen.generate(spec, row0, nrows)
                drv_ov = EngineSolver(prob_ov.solver, row0, cfg["m"], world, rank)
            eff = int(prob_ov.solver.get_option("shard_overlap"))
            if eff != 1 and world > 1:
                raise RuntimeError("the engine could not set the overlapped schedule up (shard_overlap = 0)")
        except Exception as e:  # noqa: BLE001
            ok, err = False, repr(e)
        if all_ranks_ok(ok, dist, torch):
            measure("overlap", drv_ov, prob_ov, {"LSQRHIP_SHARD_OVERLAP": "1"},
                    "the n-vector exchanges in parts on a stream of their own beside the products")
            if "overlap" in variants and variants["overlap"].get("validated"):
                variants["overlap"]["parts"] = int(prob_ov.solver.get_option("shard_parts"))
        else:
            variants["overlap"] = {"validated": False, "error": (err if not ok else "set-up failed on another rank")[:120]}
        del drv_ov, prob_ov
        # copy : the plain schedule with the n-vector exchanges as copy-engine pulls from IPC-mapped peer buffers
        #        (LSQRHIP_SHARD_COPY=1 at comm_init: a handle of its own) -- no RCCL send / receive kernel on the CUs
        prob_cp, drv_cp, ok = None, None, True
        state["doing"] = "copy (build + IPC handles)"
        try:
            with _Env({"LSQRHIP_SHARD_COPY": "1"}):
                prob_cp = devgen.generate(spec, row0, nrows)
                drv_cp = EngineSolver(prob_cp.solver, row0, cfg["m"], world, rank)
            if int(prob_cp.solver.get_option("shard_copy")) != 1 and world > 1:
                raise RuntimeError("the ranks could not map each other's buffers (hipIpcOpenMemHandle): shard_copy = 0")
        except Exception as e:  # noqa: BLE001
            ok, err = False, repr(e)
        if all_ranks_ok(ok, dist, torch):
            measure("copy", drv_cp, prob_cp, {},
                    "the plain schedule, n-vector exchanges as copy-engine pulls over IPC-mapped buffers (no RCCL send / receive kernels)")
        else:
            variants["copy"] = {"validated": False, "error": (err if not ok else "set-up failed on another rank")[:120]}
        del drv_cp, prob_cp
        # overlap_copy : the overlapped schedule with its parts as copy-engine pulls (fences: 8-byte all-gathers on the
        #        second communicator) -- the exchanges beside the products AND no send / receive kernel on the CUs
        prob_oc, drv_oc, ok = None, None, True
        state["doing"] = "overlap_copy (build + communicator + IPC handles)"
        try:
            with _Env({"LSQRHIP_SHARD_OVERLAP": "1", "LSQRHIP_SHARD_WORLD": str(world), "LSQRHIP_SHARD_COPY": "1"}):
                prob_oc = devgen.generate(spec, row0, nrows)
                drv_oc = EngineSolver(prob_oc.solver, row0, cfg["m"], world, rank)
            if world > 1 and (int(prob_oc.solver.get_option("shard_copy")) != 1 or
                              int(prob_oc.solver.get_option("shard_overlap")) != 1):
                raise RuntimeError("overlap + copy could not be set up (shard_overlap / shard_copy = 0)")
        except Exception as e:  # noqa: BLE001
            ok, err = False, repr(e)
        if all_ranks_ok(ok, dist, torch):
            measure("overlap_copy", drv_oc, prob_oc, {"LSQRHIP_SHARD_OVERLAP": "1"},
                    "exchanges in parts beside the products, as copy-engine pulls over IPC-mapped buffers")
        else:
            variants["overlap_copy"] = {"validated": False, "error": (err if not ok else "set-up failed on another rank")[:120]}
        del drv_oc, prob_oc
        if rank == 0:
            best = max((k for k, v in variants.items() if v.get("validated")), key=lambda k: variants[k]["value"])
            out["value"] = variants[best]["value"]
            out["ms_per_step"] = variants[best]["ms_per_step"]
            out["config"]["schedule"] = best
            out["overlap"] = 1 if best in ("overlap", "overlap_copy") else 0
            out["copy"] = 1 if best in ("copy", "overlap_copy") else 0
            if best in results:
                out["result"] = results[best]
            if out.get("value_1gpu_same_workload"):
                out["speedup_vs_1gpu_same_workload"] = out["value"] / out["value_1gpu_same_workload"]
            # the one-process form (Fortran `ngpu = N`) on this node's devices, small system, untimed
            state["doing"] = "inprocess_sharded_check"
            try:
                chk = inprocess_sharded_check(world)
            except Exception as e:  # noqa: BLE001
                chk = {"ok": False, "error": repr(e)}
            detail["inprocess_sharded_check"] = chk
            out["inprocess_sharded_check"] = {k: (v[:120] if isinstance(v, str) else v) for k, v in chk.items()
                                              if k in ("ok", "ngpu", "rel_dx", "skipped", "error")}
        wd.cancel()
    if rank == 0:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        _bench.emit(out, detail, args.detail)
    dist.barrier()
    dist.destroy_process_group()
