#!/usr/bin/env python3
"""bench.py -- LSQR iterations/s + aprod SpMV GB/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...                     # starts its N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE LSQR iteration (mode-1 SpMV + mode-2 SpMV + x/w update + the scalar recurrences) over the whole
system.  The timed region is one `solve` of exactly K iterations (atol = btol = conlim = 0, itnlim = K -> istop = 5) with
the matrix, b and x already resident in HBM; W warm-up iterations run first as a separate solve.

Workload, at EVERY N: BASELINE.json configs[3] made concrete per SURVEY.md 8d -- m = n = 10^7 random sparse least
squares, 100 nonzeros per row (10^9 nonzeros), damp = 1e-3 -- whole on the one GPU at N = 1, row-block sharded at N > 1
(lsqr_amd/dist_bench.py): total work fixed as N grows, "scaling": "strong", and the lines of a series divide directly.

The LAST line of stdout is the ONE JSON line (rank 0), at most 4 KB (tests/test_bench_line_keys.py):
  contract keys, `result`;
  roofline      the dominant kernel, k_spmv_csb of aprod mode 1: `achieved` = SURVEY 8d's algorithmic bytes B1 =
                12 nnz + P (m + 1) + 8 n + 16 m  /  the average duration of a product (HIP events on the solver's stream
                around back-to-back launches), `frac` = that / 8 TB/s; `traffic` = HBM bytes per product from rocprofv3 PMC
                passes of THIS build made by this run (FETCH_SIZE and WRITE_SIZE in passes of their own, FETCH_SIZE x2
                on gfx950); `frac_mode2` the same figure for the transposed product;
  cpu_baseline  the reference's own CPU path (oracle/_ref), 1 core, on a scaled-down instance of the same generator;
  configs[]     {workload, it_s, frac_mode1, frac_mode2} for the other BASELINE configurations one GPU holds and the
                row blocks a rank of eight holds -- measured by a CHILD process after the headline (a failure or a
                time-out there costs its entries, never the line);
  detail        path of the side file with everything else (per-kernel times, layouts, physical bytes, PMC detail).
`--workload SPEC --extras off` measures one workload alone; `--roofline-only` only its mode-1 product (what a
`rocprofv3 --kernel-trace --stats` of the same command averages: profiles/rNN/*_roofline.txt).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured float4 copy)
INFINITY_CACHE = 256 << 20
HEADLINE = "random:10000000:10000000:100"      # BASELINE.json configs[3] (100 per row), = dist_bench.DEFAULT_SPEC
HEADLINE_NOTE = "BASELINE.json configs[3]: 10M x 10M random"
LINE_MAX = 4096            # bytes of the one JSON line (the driver's reader gave up on round 5's 22 KB)
PRODUCT_KERNELS = ("k_spmv_", "k_panel_combine", "k_csb_combine", "k_csb_xmax")
# The other BASELINE configurations one GPU holds and the shapes an 8-GPU run executes, each measured by the same run in
# a child process: (spec, timed iterations, warm-up iterations, tag, engine) -- `engine`: also the sharded engine at world
# 1 on that block (what one rank of eight does per iteration, exchanges apart).
COMPACT_CONFIGS = [
    ("poisson2d:1000:1000", 20, 5, "configs[1]", False),
    ("random:4000000:1000000:1000", 6, 2, "configs[2] literal", False),
    ("powerlaw:5000000:2000000:10000", 40, 4, "configs[4]", False),
    ("random:1250000:10000000:100", 60, 6, "configs[3] rank block N=8", True),
    ("random:1250000:10000000:1000", 10, 2, "configs[3] rank block N=8 at r=1000", False),
]
CHILD_TIMEOUT = 240        # seconds for the configs[] child (it takes ~20)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed LSQR iterations (default 200)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed iterations first (default steps / 10)")
    ap.add_argument("--workload", default="auto",
                    help="auto | poisson2d:NX:NY | mesh2d:NX:NY:BX:BY | random:M:N:PER_ROW | powerlaw:M:N:DMAX")
    ap.add_argument("--cpu-iters", type=int, default=20, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--extras", choices=["on", "off"], default="on",
                    help="off: only the one workload (no configs[] child)")
    ap.add_argument("--configs", choices=["on", "off"], default="on", help="off: skip the configs[] child (~1.5 min)")
    ap.add_argument("--traffic", choices=["live", "off"], default="live",
                    help="live: HBM bytes per launch from rocprofv3 PMC passes run as child processes")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where everything that does not fit the line goes")
    ap.add_argument("--roofline-only", action="store_true",
                    help="build the workload and time ONLY its products (what a `rocprofv3 --kernel-trace "
                         "--stats` of this command then averages: profiles/rNN/*_roofline.txt)")
    ap.add_argument("--roofline-mode", type=int, choices=[1, 2], default=1, help="--roofline-only: which product (aprod mode)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--configs-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.steps is None:        # ~5 ms per iteration on one GPU
        a.steps = 200
    if a.warmup is None:
        a.warmup = max(1, a.steps // 10)
    return a


def make_problem(spec: str):
    from lsqr_amd import problems as P
    kind, *a = spec.split(":")
    if kind == "poisson2d":
        return P.poisson2d(int(a[0]), int(a[1]))
    if kind == "mesh2d":
        return P.mesh2d(int(a[0]), int(a[1]), int(a[2]), int(a[3]))
    if kind == "random":
        return P.random_rows(int(a[0]), int(a[1]), int(a[2]), damp=1e-3)
    if kind == "powerlaw":
        return P.powerlaw_rows(int(a[0]), int(a[1]), dmax=int(a[2]))
    raise SystemExit(f"unknown workload {spec}")


def cpu_baseline(p, iters: int):
    """The reference's CPU path on the same system, 1 core (it has no threading)."""
    import oracle
    rf = oracle.ref()
    eng, kind = (rf, "reference") if rf is not None else (oracle.port(), "port")
    t0 = time.perf_counter()
    r = eng.solve(p.m, p.n, p.irow, p.icol, p.a, p.b, damp=p.damp, itnlim=iters)
    dt = time.perf_counter() - t0
    assert r.itn == iters, (r.itn, r.istop)
    return {"value": r.itn / dt, "unit": "it/s", "cores": 1, "kind": kind,
            "sample": f"{p.name}: same (irow,icol,a,b), {iters} iterations in {dt:.2f} s, "
                      f"{'oracle/_ref (reference compiled with amdflang -O2)' if kind == 'reference' else 'oracle C port'}"}


# ---------------------------------------------------------------------------------------------
# one workload on this GPU
# ---------------------------------------------------------------------------------------------
class _Env:
    """LSQRHIP_* knobs for the duration of one build (the library reads them at create)."""

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


def build_workload(spec: str, env=None, itnlim=100, rows=None):
    """(solver, d_b, facts) with the matrix resident in HBM; small systems also keep the host copy.
    `rows` = (row0, nrows): only that row block of the system (one rank's share of a sharded solve)."""
    from lsqr_amd import capi, devgen
    from lsqr_amd.solver import lsqr_solver_ez
    cfg = devgen.parse_spec(spec)
    nnz_est = cfg["m"] * (5 if cfg["kind"] in ("poisson2d", "mesh2d") else cfg.get("per_row", 30))
    host = None
    with _Env(env):
        if rows is not None:
            dp = devgen.generate(spec, int(rows[0]), int(rows[1]), itnlim=itnlim)
            s, d_b = dp.solver, dp.d_b
            facts = dict(name=f"{spec} rows {rows[0]}..{rows[0] + rows[1]}", m=dp.nrows, n=dp.n, nnz=dp.nnz, damp=dp.damp, spec="")
        elif nnz_est <= 60_000_000 and not env:     # the host copy only exists for the CPU baseline
            host = make_problem(spec)
            s = lsqr_solver_ez().initialize(host.m, host.n, host.a, host.irow, host.icol, itnlim=itnlim)
            d_b = capi.DeviceBuffer.from_array(host.b)
            facts = dict(name=host.name, m=host.m, n=host.n, nnz=host.nnz, damp=host.damp, spec=spec)
        else:                                        # generated in HBM (bit-identical generator, csrc/gen_api.h)
            dp = devgen.generate(spec, itnlim=itnlim)
            s, d_b = dp.solver, dp.d_b
            facts = dict(name=spec, m=dp.m, n=dp.n, nnz=dp.nnz, damp=dp.damp, spec=spec)
    return s, d_b, facts, host


def describe_layout(info):
    if info["sell"] == 4:
        return "structure patterns (one byte per row + 8-byte values, column offsets in LDS)", "k_spmv_spat"
    pair = " -- paired rows" if info.get("pat_pair") else ""
    if info["sell"] == 3 and info.get("pat_wide"):
        return "row patterns (two bytes per row, the table through L2)" + pair, "k_spmv_pat2p" if pair else "k_spmv_pat2"
    if info["sell"] == 3:
        return "row patterns (one byte per row, patterns in LDS)" + pair, "k_spmv_patp" if pair else "k_spmv_pat"
    if info["sell"] == 2:
        return "sliced ELL, packed 16-byte records", "k_spmv_sellp"
    if info["sell"]:
        return "sliced ELL", "k_spmv_sell"
    if info["xlds"] == 3:
        return "column-swept row blocks (LDS accumulators)", "k_spmv_csb"
    if info["xlds"] == 2:
        return "LDS column panels (wave windows)", "k_spmv_xlw + k_panel_combine"
    if info["panels"] > 1:
        return "L2 column panels", "k_spmv_fused + k_panel_combine"
    return "row windows", "k_spmv_fused"


def product_roofline(s, facts, reps, traffic=None):
    """(`roofline` object of the mode-1 product, detail): average of `reps` back-to-back launches inside ONE HIP event
    pair on the solver's stream (what rocprofv3's kernel trace reports as the kernel's average duration x the launches
    per product).  `achieved` / `frac` are on SURVEY 8d's ALGORITHMIC bytes (B1: 8-byte values, 4-byte columns, row
    pointers, x once, y read and written); the bytes of the layout actually stored are beside it (`frac_layout`).  A
    layout that compresses (row patterns, dictionaries) on a cache-resident system can exceed 1: `bound` then says
    "cache" -- such a product is not an HBM stream of B1."""
    info = s.info()
    m, n, nnz = facts["m"], facts["n"], facts["nnz"]
    P = info["rowptr_bytes"]
    alg1 = 12 * nnz + P * (m + 1) + 8 * n + 16 * m            # B1
    alg2 = 12 * nnz + P * (n + 1) + 8 * m + 16 * n            # B2
    lay1 = info["csr_bytes"] + 8 * n + 16 * m                  # matrix as stored + x once + y read and written
    lay2 = info["csrt_bytes"] + 8 * m + 16 * n
    avg1, avg2, avg3 = (s.bench_kernel(w, reps) for w in (1, 2, 3))
    layout, kname = describe_layout(info)
    ach = alg1 / (avg1 * 1e-3) / 1e9
    frac = ach / HBM_PEAK_GBS
    ach2 = alg2 / (avg2 * 1e-3) / 1e9
    wset = info["csr_bytes"] + info["csrt_bytes"] + 8 * (m + 4 * n)
    lpp = s.get_option("launches_mode1")
    roof = {"bound": "cache" if frac > 1.0 else "hbm", "kernel": f"{kname} (aprod mode 1)", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
            "traffic": traffic.get("bytes_per_launch") if traffic else None,
            "bytes_per_launch": alg1, "bytes_are": "SURVEY 8d B1 = 12 nnz + P (m + 1) + 8 n + 16 m (algorithmic)",
            "avg_launch_us": avg1 * 1e3, "launches": reps,
            "kernel_launches_per_product": lpp,        # > 1: a product is that many launches of the kernel (csb.h: one
            "avg_kernel_launch_us": avg1 * 1e3 / lpp,  # per round of row blocks); rocprofv3's per-kernel average is this
            "frac_layout": lay1 / (avg1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "layout_bytes_per_launch": lay1,
            "frac_mode2": ach2 / HBM_PEAK_GBS, "avg_launch_us_mode2": avg2 * 1e3}
    detail = {
        "layout": layout,
        "format": ({"row_bytes": 2 if info.get("pat_wide") else 1, "value_bytes": 0, "col_bytes": 0}
                   if info["sell"] == 3 else
                   {"value_bytes": info["value_bytes"], "col_bytes": info["col_bytes"], "dict_entries": info["dict_entries"]}),
        "resident": ("infinity cache (iteration working set %.0f MB < 256 MB)" % (wset / 1e6)) if wset < INFINITY_CACHE else
                    "hbm (iteration working set %.1f GB)" % (wset / 1e9),
        "traffic_detail": traffic,
        "kernels": {
            "spmv_mode1": {"avg_launch_us": avg1 * 1e3, "survey8d_bytes": alg1, "layout_bytes": lay1,
                           "frac_survey8d": frac, "frac_layout": roof["frac_layout"]},
            "spmv_mode2": {"avg_launch_us": avg2 * 1e3, "survey8d_bytes": alg2, "layout_bytes": lay2,
                           "frac_survey8d": ach2 / HBM_PEAK_GBS, "frac_layout": lay2 / (avg2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "kernel_launches_per_product": s.get_option("launches_mode2")},
            "update_xw": {"avg_launch_us": avg3 * 1e3, "bytes_per_launch": 40 * n,
                          "frac": 40 * n / (avg3 * 1e-3) / 1e9 / HBM_PEAK_GBS},
        },
        "iter_bytes_survey8d": alg1 + alg2 + 40 * n, "iter_bytes_layout": lay1 + lay2 + 40 * n,
    }
    if info["xlds"] == 3:
        detail["csb_lockstep"] = s.get_option("csb_lockstep_mode1")   # chunks per wave and lock-step step (0: free-running sweep)
    return roof, detail


def timed_solve(s, d_b, d_x, damp, K):
    """EXACTLY K iterations.  With atol = btol = conlim = 0 the default workloads never stop before
    the limit; a well-conditioned --workload can reach machine precision first: the solve is then
    started again on the same b until K iterations have run (`restarts`)."""
    import torch
    done, restarts = 0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while done < K:
        s.itnlim = K - done
        r = s.solve_device(d_b.ptr.value, d_x.ptr.value, damp)
        done += r.itn
        if done < K:
            restarts += 1
            if r.itn == 0:
                raise SystemExit("bench.py: workload stops at iteration 0 (b = 0?)")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert done == K and (restarts > 0 or r.istop == 5), (done, r.itn, r.istop)
    return dt, r, restarts


def roofline_only(args):
    """Only ONE product kernel: 3 warm + `reps` back-to-back launches of mode 1 (or, `--roofline-mode 2`, of mode 2).  Under
    `rocprofv3 --kernel-trace --stats` the kernel's average duration x kernel_launches_per_product is avg_launch_us of the
    line (nothing else of that kernel's name runs in the process)."""
    spec = HEADLINE if args.workload == "auto" else args.workload
    mode = args.roofline_mode
    s, d_b, facts, _ = build_workload(spec, None)
    info = s.info()
    m, n, nnz, P = facts["m"], facts["n"], facts["nnz"], info["rowptr_bytes"]
    reps = 200 if nnz < 50_000_000 else (50 if nnz < 200_000_000 else 10)
    avg = s.bench_kernel(mode, reps)
    alg = 12 * nnz + (P * (m + 1) + 8 * n + 16 * m if mode == 1 else P * (n + 1) + 8 * m + 16 * n)
    lay = (info["csr_bytes"] + 8 * n + 16 * m) if mode == 1 else (info["csrt_bytes"] + 8 * m + 16 * n)
    lpp = s.get_option(f"launches_mode{mode}")
    ach = alg / (avg * 1e-3) / 1e9
    print(json.dumps({"workload": spec, "env": {k: v for k, v in os.environ.items() if k.startswith("LSQRHIP_")},
                      "roofline": {"kernel": describe_layout(info)[1] + f" (aprod mode {mode})", "bytes_per_launch": alg,
                                   "bytes_are": f"SURVEY 8d B{mode} (algorithmic)", "avg_launch_us": avg * 1e3,
                                   "kernel_launches_per_product": lpp, "avg_kernel_launch_us": avg * 1e3 / lpp,
                                   "launches": reps, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac_layout": lay / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "layout_bytes_per_launch": lay}}), flush=True)


def graph_batch(K):
    """a divisor of K when there is a good one (no predicated-off tail iterations in the timed solve), else up to 50
    iterations (launches past the stop are ~us-scale no-ops)"""
    return next((g for g in (100, 50, 40, 32, 26, 20, 16) if K % g == 0), min(50, K + (K & 1)))


def measure_workload(spec, K, W, traffic=None, env=None, note="", engine=False):
    """One workload on this GPU: build (generated in HBM), warm up, W + K timed iterations, the products' roofline;
    `engine`: also the sharded engine at world 1 on it.  (full record for the detail file)"""
    from lsqr_amd import capi
    t_build = time.perf_counter()
    s, d_b, facts, _ = build_workload(spec, env, itnlim=K)
    t_build = time.perf_counter() - t_build
    d_x = capi.DeviceBuffer(8 * max(facts["n"], 1))
    s.atol = s.btol = s.conlim = 0.0
    gi = graph_batch(K)
    s.set_option("graph_iters", gi)
    # setup, untimed: the graphs of this batch size are captured and instantiated by the first solve, and the
    # interpreter and the device clocks settle over a few more (a 20-iteration solve of configs[1] is 0.5 ms of device
    # work: the first timed call after an idle start otherwise measures the ramp): at least 2 solves and 60 ms of them
    t_warm, n_warm = time.perf_counter(), 0
    while n_warm < 2 or (time.perf_counter() - t_warm < 0.06 and n_warm < 400):
        timed_solve(s, d_b, d_x, facts["damp"], min(K, 100))
        n_warm += 1
    if W > 0:                                   # the W warm-up steps of the contract, through the timed path
        timed_solve(s, d_b, d_x, facts["damp"], W)
    dt, r, restarts = timed_solve(s, d_b, d_x, facts["damp"], K)
    reps = 500 if facts["nnz"] < 50_000_000 else (100 if facts["nnz"] < 200_000_000 else 10)
    roof, detail = product_roofline(s, facts, reps, traffic)
    out = {"workload": f"{spec} m={facts['m']} n={facts['n']} nnz={facts['nnz']} damp={facts['damp']}", "spec": spec,
           "note": note, "env": env or {}, "n_gpus": 1, "steps": K, "warmup": W, "value": K / dt, "unit": "it/s",
           "ms_per_step": 1e3 * dt / K, "restarts": restarts, "graph_iters": gi, "generate_and_build_s": t_build,
           "result": {"istop": r.istop, "itn": r.itn, "anorm": r.anorm, "rnorm": r.rnorm},
           "iter_gbps_survey8d": detail["iter_bytes_survey8d"] * K / dt / 1e9,
           "roofline": roof, "detail": detail}
    if engine:      # what ONE rank of an N-GPU run does per iteration, exchanges apart (csrc/shard_engine.h at world 1)
        try:
            import torch

            from lsqr_amd.dist import EngineSolver
            drv = EngineSolver(s, 0, facts["m"], 1, 0)
            kw = dict(damp=facts["damp"], atol=0.0, btol=0.0, conlim=0.0)
            drv.solve(d_b.ptr.value, itnlim=max(W, 2), **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            re = drv.solve(d_b.ptr.value, itnlim=K, **kw)
            torch.cuda.synchronize()
            out["engine_world1_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / max(re.itn, 1)
            out["engine_world1_itn"] = re.itn
        except Exception as e:      # noqa: BLE001
            out["engine_world1_error"] = repr(e)
    del s, d_b, d_x
    return out


def compact_entry(full, tag):
    """what the one line carries of a configs[] record"""
    if "error" in full:
        return {"workload": full.get("spec", "?"), "tag": tag, "error": full["error"][:80]}
    r = full["roofline"]
    e = {"workload": full["spec"], "tag": tag, "steps": full["steps"], "it_s": round(full["value"], 2),
         "frac_mode1": round(r["frac"], 4), "frac_mode2": round(r["frac_mode2"], 4),
         "us_mode1": round(r["avg_launch_us"], 2), "us_mode2": round(r["avg_launch_us_mode2"], 2)}
    if r["bound"] != "hbm":
        e["bound"] = r["bound"]
    if "engine_world1_ms_per_step" in full:
        e["engine_world1_ms_per_step"] = round(full["engine_world1_ms_per_step"], 4)
    return e


def configs_child():
    """`bench.py --configs-child`: the COMPACT_CONFIGS one after the other, each guarded; prints ONE line
    `CONFIGS_JSON [...]` of full records (the parent keeps the compact form in its line, the rest in the detail file)."""
    import torch
    from lsqr_amd import capi
    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible; the HIP path has no CPU fallback")
    t0 = time.perf_counter()
    recs = []
    for spec, K, W, tag, engine in COMPACT_CONFIGS:
        if time.perf_counter() - t0 > CHILD_TIMEOUT - 90:
            recs.append({"spec": spec, "tag": tag, "error": "skipped: the child's time budget was used up"})
            continue
        try:
            rec = measure_workload(spec, K, W, note=tag, engine=engine)
        except (Exception, SystemExit) as e:      # noqa: BLE001  (one entry's failure costs that entry)
            rec = {"spec": spec, "error": repr(e)}
        rec["tag"] = tag
        recs.append(rec)
        print("CONFIGS_PARTIAL " + json.dumps(rec), flush=True)
    print("CONFIGS_JSON " + json.dumps(recs), flush=True)


def run_configs_child():
    """(records, note): the configs[] child, its time bounded; whatever it finished is kept."""
    env = dict(os.environ)
    try:
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--configs-child"], capture_output=True,
                            text=True, timeout=CHILD_TIMEOUT, env=env, cwd=ROOT)
        out, note = cp.stdout, (None if cp.returncode == 0 else f"configs child rc {cp.returncode}: {cp.stderr[-200:]}")
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        note = f"configs child timed out after {CHILD_TIMEOUT} s"
    full = [ln for ln in out.splitlines() if ln.startswith("CONFIGS_JSON ")]
    if full:
        return json.loads(full[-1][len("CONFIGS_JSON "):]), note
    return [json.loads(ln[len("CONFIGS_PARTIAL "):]) for ln in out.splitlines() if ln.startswith("CONFIGS_PARTIAL ")], note


def emit(line: dict, detail: dict, path: str):
    """The detail file first, then the ONE line -- at most LINE_MAX bytes: optional keys go, in this order, until it
    fits (the contract keys, `roofline` and `cpu_baseline` always stay)."""
    try:
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        line["detail"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        line["detail"] = f"not written: {e!r}"
    text = json.dumps(line)
    for victim in ("notes", "variants", "configs", "result"):
        if len(text) <= LINE_MAX:
            break
        line.pop(victim, None)
        line["dropped_for_length"] = line.get("dropped_for_length", []) + [victim]
        text = json.dumps(line)
    assert len(text) <= LINE_MAX, len(text)
    sys.stdout.flush()
    print(text, flush=True)


# ---------------------------------------------------------------------------------------------
# HBM traffic per launch: rocprofv3 PMC passes of this build, as child processes
# ---------------------------------------------------------------------------------------------
def pmc_child(counter: str):
    """Runs under `rocprofv3 --pmc COUNTER`: builds each workload of the plan (stdin: JSON list of
    [spec, env] or [spec, env, row0, nrows]) and launches its mode-1 product 3 + reps times; prints the
    launch manifest."""
    plan = json.loads(sys.stdin.read())
    manifest = []
    for spec, env, *rows in plan:
        s, d_b, facts, _ = build_workload(spec, env, rows=rows or None)
        reps = 10 if facts["nnz"] < 200_000_000 else 4
        s.bench_kernel(1, reps)
        manifest.append({"spec": spec, "env": env, "rows": rows, "launches": 3 + reps,
                         "kernels_per_product": s.get_option("dispatches_mode1")})
        del s, d_b
    print("PMC_MANIFEST " + json.dumps(manifest), flush=True)


def live_traffic(plan, timeout=180):
    """{(spec, env-json): {bytes_per_launch, fetch_bytes, write_bytes}}.  FETCH_SIZE and WRITE_SIZE are
    collected in passes of their own (TCC slots); FETCH_SIZE is doubled (gfx950 tallies 128-byte
    requests at 64 bytes, MI355X_MICROARCH.md "HBM").  Returns ({}, reason) when rocprofv3 cannot run."""
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return {}, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {}, "this run is itself being profiled; PMC passes skipped"
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCPROF", "ROCP_", "HSA_TOOLS"))}
    env["TMPDIR"] = "/tmp"
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="lsqr_pmc_", dir="/tmp")
        try:
            cp = subprocess.run([rp, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                                 sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", counter],
                                input=json.dumps(plan), capture_output=True, text=True, timeout=timeout, cwd="/tmp",
                                env=env)
            man = [ln for ln in cp.stdout.splitlines() if ln.startswith("PMC_MANIFEST ")]
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if cp.returncode != 0 or not man or not files:
                return {}, f"PMC pass {counter} failed (rc {cp.returncode}): {cp.stderr[-300:]}"
            manifest = json.loads(man[0][len("PMC_MANIFEST "):])
            rows = []
            for f in files:
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == counter and any(k in r.get("Kernel_Name", "") for k in PRODUCT_KERNELS):
                        rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            rows.sort()
            pos = 0
            for mf in manifest:       # dispatch order = plan order; the build launches no product kernel
                nk = mf["launches"] * mf["kernels_per_product"]
                vals = [v for _, v in rows[pos:pos + nk]]
                pos += nk
                if len(vals) != nk:
                    return {}, f"PMC pass {counter}: {len(rows)} product dispatches, manifest wants more"
                live = vals[3 * mf["kernels_per_product"]:]          # skip the 3 warm launches
                per_launch = sum(live) / (len(live) / mf["kernels_per_product"])
                sums.setdefault((mf["spec"], json.dumps(mf["env"], sort_keys=True), tuple(mf.get("rows") or ())),
                                {})[counter] = per_launch
            if pos != len(rows):
                return {}, f"PMC pass {counter}: {len(rows) - pos} unexpected product dispatches"
        except Exception as e:      # noqa: BLE001
            return {}, f"PMC pass {counter}: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for key, c in sums.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch, write = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0     # counters are in KB
            out[key] = {"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes made by this run (same build), "
                                  "FETCH_SIZE x2 (gfx950), KB -> bytes; mean over the live launches"}
    return out, None


# ---------------------------------------------------------------------------------------------
def run_single(args):
    spec = HEADLINE if args.workload == "auto" else args.workload
    extras = args.extras == "on" and spec == HEADLINE
    t_run = time.perf_counter()
    notes = []

    # PMC passes first, as children, before this process touches the GPU
    traffic, traffic_note = ({}, "--traffic off")
    if args.traffic == "live" and not args.no_roofline:
        traffic, traffic_note = live_traffic([[spec, {}]])
    tr = traffic.get((spec, "{}", ()))

    import torch
    from lsqr_amd import capi, devgen
    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible; the HIP path has no CPU fallback")
    K, W = args.steps, args.warmup
    full = measure_workload(spec, K, W, tr, note=HEADLINE_NOTE if spec == HEADLINE else "")
    cfgname = f" ({HEADLINE_NOTE}, whole on one GPU)" if spec == HEADLINE else ""
    roof = dict(full["roofline"])
    if roof["traffic"] is None:
        roof["traffic_note"] = (traffic_note or "no PMC rows for this workload")[:160]
    if args.no_roofline:
        roof = None
    line = {
        "metric": "lsqr_iterations_per_sec", "value": full["value"], "unit": "it/s", "n_gpus": 1,
        "steps": K, "warmup": W, "ms_per_step": full["ms_per_step"], "higher_is_better": True,
        # the same system at every N, total work fixed: the N > 1 lines (row-block sharded) divide by this one
        "scaling": "strong" if spec == HEADLINE else "n/a (one GPU, one workload)",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": full["workload"] + cfgname, "graph_iters": full["graph_iters"], "restarts": full["restarts"]},
        "result": full["result"],
        "roofline": roof,
        "spmv_gbps": {"mode1": full["roofline"]["achieved"], "mode2": full["roofline"]["frac_mode2"] * HBM_PEAK_GBS,
                      "iteration": full["iter_gbps_survey8d"], "bytes": "SURVEY 8d algorithmic (B1, B2, BI)"},
    }
    detail = {"headline": full, "traffic_note": traffic_note}

    # cpu_baseline: the reference's CPU path, 1 core, on a bounded sample -- the workload itself where the host can hold it
    cpu = None
    if args.cpu_iters > 0:
        try:
            cfg = devgen.parse_spec(spec)
            if cfg["kind"] == "random" and full_nnz(full) > 60_000_000:
                from lsqr_amd.dist_bench import cpu_baseline_scaled
                cpu = cpu_baseline_scaled(spec, full_nnz(full), iters=args.cpu_iters)
            else:
                host = make_problem(spec)
                if host.nnz > 60_000_000:
                    raise RuntimeError("no scaled-down generator for this workload")
                cpu = cpu_baseline(host, max(args.cpu_iters, 200 if host.nnz < 10_000_000 else args.cpu_iters))
                del host
        except Exception as e:      # noqa: BLE001
            cpu = {"error": repr(e)[:200]}
    line["cpu_baseline"] = cpu

    if extras and args.configs == "on" and not args.no_roofline:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        recs, note = run_configs_child()
        if note:
            notes.append(note)
        line["configs"] = [compact_entry(r, r.get("tag", "")) for r in recs]
        detail["configs"] = recs
    if notes:
        line["notes"] = [n[:200] for n in notes]
    detail["run_seconds"] = time.perf_counter() - t_run
    emit(line, detail, args.detail)


def full_nnz(full):
    return int(full["workload"].split("nnz=")[1].split()[0])


def spawn_ranks(args):
    """`bench.py --gpus N` without a launcher: start the N ranks ourselves, as a child process (never an
    exec), BEFORE this process touches the GPU, and pass the child's JSON line through."""
    import torch
    have = torch.cuda.device_count()          # does not initialise a context
    if have < args.gpus and os.environ.get("LSQR_RANKS_SHARE_GPU", "0") in ("", "0"):   # (the test switch: dist_bench.share_one_gpu)
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node shows {have}")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cp = subprocess.run(cmd, env=env)
    raise SystemExit(cp.returncode)


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args.pmc_child)
    if args.configs_child:
        return configs_child()
    if args.roofline_only:
        return roofline_only(args)
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if args.gpus > 1 and world == 0:
        return spawn_ranks(args)
    if args.gpus > 1 or world > 1 or os.environ.get("LSQR_BENCH_FORCE_DIST") == "1":
        from lsqr_amd.dist_bench import run_distributed
        run_distributed(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
