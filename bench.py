#!/usr/bin/env python3
"""bench.py -- LSQR iterations/s + aprod SpMV GB/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...                     # starts its N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE LSQR iteration (mode-1 SpMV + mode-2 SpMV + x/w update + the scalar
recurrences) over the whole system.  The timed region is one `solve` of exactly K
iterations (atol = btol = conlim = 0, itnlim = K -> istop = 5) with the matrix, b and x
already resident in HBM; W warm-up iterations run first as a separate solve.

N = 1 workload: BASELINE.json configs[1] -- 1M x 1M 5-point Poisson (nnz 4 996 000),
damp = 0.  N > 1: the row-block sharded solve of configs[3] (lsqr_amd/dist_bench.py).

The ONE JSON line (rank 0) carries, for the dominant kernel (aprod mode 1):
  roofline          PHYSICAL: bytes of the layout the build chose / average launch time (HIP events on
                    the solver's stream), frac = that / 8 TB/s (always <= 1); `traffic` = HBM bytes per
                    launch from rocprofv3 PMC passes of THIS build made by this run (FETCH_SIZE and
                    WRITE_SIZE in passes of their own, FETCH_SIZE x2 on gfx950); `effective_gbps` = the
                    SURVEY 8d algorithmic bytes (12 B per nonzero ...) / the same time, labelled as such.
  roofline_hbm      the same kernel family on HBM-RESIDENT instances (configs[1] fits the 256 MB
                    Infinity Cache): poisson2d:4000:4000 as row patterns, packed records, sliced ELL with 8-byte values
                    and structure patterns; mesh2d:4000:4000:16:16 (the same operator with its coefficient constant on
                    each of 256 regions: 2304 distinct rows) as wide row patterns.
  strong_scaling_n1 configs[3] (10M x 10M, 1e9 nonzeros) whole on this GPU, with its own roofline:
                    N = 1 of the series the --gpus N lines continue.
  roofline_configs  the same object for configs[2] at its literal 1000 per row, configs[4] and the r = 1000 rank block.
  cpu_baseline      the reference's own CPU path (oracle/_ref), 1 core, bounded sample.
`--workload SPEC --extras off` measures one workload alone (what profiles/r02/*.txt were made with).
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
HEADLINE = "poisson2d:1000:1000"
HBM_INSTANCES = [("poisson2d:4000:4000", {}, "row patterns (what the build chooses for a constant-coefficient stencil)"),
                 ("poisson2d:4000:4000", {"LSQRHIP_PAT": "0"}, "packed records: value dictionary (1-byte codes), 16-bit columns"),
                 ("poisson2d:4000:4000", {"LSQRHIP_PAT": "0", "LSQRHIP_VAL8": "0", "LSQRHIP_SPAT": "0"},
                  "sliced ELL: 8-byte values, 16-bit columns (matrices without a repeating structure)"),
                 ("poisson2d:4000:4000", {"LSQRHIP_PAT": "0", "LSQRHIP_VAL8": "0"},
                  "structure patterns: 8-byte values, no column indices (what a variable-coefficient stencil gets)"),
                 ("mesh2d:4000:4000:16:16", {},
                  "wide row patterns: the same five-point operator with its coefficient constant on each of 16 x 16 regions "
                  "(2304 distinct rows): two bytes per row, the table through L1 / L2 (round 4: structure patterns)")]
PRODUCT_KERNELS = ("k_spmv_", "k_panel_combine", "k_csb_combine", "k_csb_xmax")
# the N = 1 line measures configs[1]; the series the --gpus N lines continue is strong_scaling_n1 (configs[3])
SCALING_N1 = "n/a (configs[1]; the strong-scaling series is strong_scaling_n1)"
# (Round 4 pasted builder-measured "ceilings of the access pattern" -- scripts/csb_ceiling.hip -- into this line as
# constants.  Round 5's lock-step sweep runs PAST them: they were ceilings of one schedule, not of the memory system, and
# are gone from the line.  What is left is measured by the run itself.)
# The other BASELINE configurations one GPU holds, each with the roofline object of its mode-1 product measured by THIS
# run (`roofline_configs`): configs[2] at its literal 1000 per row, configs[4], and the block one rank of eight holds of
# configs[3] at SURVEY 8d's r = 1000.  (spec, solve iterations, note)
ROOFLINE_CONFIGS = [
    ("random:4000000:1000000:1000", 6, "BASELINE.json configs[2] at its literal size: 4M x 1M, 1000 per row (4e9 nonzeros, 96 GB of layouts)"),
    ("powerlaw:5000000:2000000:10000", 40, "BASELINE.json configs[4]: power-law rows up to 10^4, 5M x 2M"),
    ("random:1250000:10000000:1000", 10, "one rank's block (N = 8) of configs[3] at SURVEY 8d's r = 1000: 1.25M x 10M, 1.25e9 nonzeros"),
]
GENERAL_INSTANCE = 2      # HBM_INSTANCES[2]: sliced ELL with 8-byte values -- the best HBM-resident GENERAL short-row kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed LSQR iterations (default 2000; 400 for --gpus > 1)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed iterations first (default steps / 10)")
    ap.add_argument("--workload", default="auto",
                    help="auto | poisson2d:NX:NY | mesh2d:NX:NY:BX:BY | random:M:N:PER_ROW | powerlaw:M:N:DMAX")
    ap.add_argument("--cpu-iters", type=int, default=1000, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--extras", choices=["on", "off"], default="on",
                    help="off: only the one workload (no HBM-resident instances, no configs[3] point)")
    ap.add_argument("--configs", choices=["on", "off"], default="on",
                    help="off: skip roofline_configs (configs[2] literal, configs[4], the r = 1000 rank block: ~1 min of builds)")
    ap.add_argument("--no-scaling-ref", action="store_true",
                    help="skip the N = 1 point of the multi-GPU series (configs[3] whole on this GPU)")
    ap.add_argument("--traffic", choices=["live", "off"], default="live",
                    help="live: HBM bytes per launch from rocprofv3 PMC passes run as child processes")
    ap.add_argument("--roofline-only", action="store_true",
                    help="build the workload and time ONLY its mode-1 product (what a `rocprofv3 --kernel-trace "
                         "--stats` of this command then averages: profiles/r02/*_roofline.txt)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    multi = a.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1
    if a.steps is None:        # N = 1: ~26 us per iteration; N > 1 (10^9 nonzeros): milliseconds
        a.steps = 400 if multi else 2000
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
    """The `roofline` object of the mode-1 product of a built workload: average of `reps` back-to-back
    launches inside ONE HIP event pair on the solver's stream (what rocprofv3's kernel trace reports as
    the kernel's average duration), against the bytes of the layout in use."""
    info = s.info()
    m, n, nnz = facts["m"], facts["n"], facts["nnz"]
    P = info["rowptr_bytes"]
    alg1 = 12 * nnz + P * (m + 1) + 8 * n + 16 * m            # B1
    alg2 = 12 * nnz + P * (n + 1) + 8 * m + 16 * n            # B2
    lay1 = info["csr_bytes"] + 8 * n + 16 * m                  # matrix as stored + x once + y read and written
    lay2 = info["csrt_bytes"] + 8 * m + 16 * n
    avg1, avg2, avg3 = (s.bench_kernel(w, reps) for w in (1, 2, 3))
    layout, kname = describe_layout(info)
    ach = lay1 / (avg1 * 1e-3) / 1e9
    frac = ach / HBM_PEAK_GBS
    wset = info["csr_bytes"] + info["csrt_bytes"] + 8 * (m + 4 * n)
    lpp = s.get_option("launches_mode1")
    roof = {"bound": "hbm", "kernel": f"{kname} (aprod mode 1)", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": frac, "traffic": None, "bytes_per_launch": lay1,
            "avg_launch_us": avg1 * 1e3, "launches": reps,
            "kernel_launches_per_product": lpp,      # > 1: a product is that many launches of the kernel (csb.h: one
            "avg_kernel_launch_us": avg1 * 1e3 / lpp,  # per round of row blocks); rocprofv3's per-kernel average is this
            "bytes_are": "the layout in use: matrix as stored + x once + y read and written (physical)",
            "effective_gbps": alg1 / (avg1 * 1e-3) / 1e9, "effective_bytes_per_launch": alg1,
            "effective_is": "SURVEY 8d algorithmic bytes (8-byte values, 4-byte columns, row pointers) / the same time; "
                            "exceeds the physical rate whenever the layout compresses -- not a roofline fraction",
            "format": ({"layout": layout, "row_bytes": 1, "value_bytes": 0, "col_bytes": 0,
                        "note": "one pattern number per row; the distinct rows (<= 256, <= 1024 entries) live in LDS"}
                       if info["sell"] == 3 else
                       {"layout": layout, "value_bytes": info["value_bytes"], "col_bytes": info["col_bytes"],
                        "dict_entries": info["dict_entries"]}),
            "resident": ("infinity cache (iteration working set %.0f MB < 256 MB: 'HBM' bytes are fabric requests "
                         "that may be served on-die)" % (wset / 1e6)) if wset < INFINITY_CACHE else
                        "hbm (iteration working set %.1f GB)" % (wset / 1e9)}
    if info["sell"] == 3 and avg1 * 1e3 < 20.0:
        roof["frac_is"] = ("a launch-latency figure: this launch lives %.1f us and moves %.0f MB -- a chain of dependent round "
                           "trips (DESIGN.md 3.4b), not a stream; the same kernel on an HBM-resident instance is "
                           "roofline_hbm[0], the layout it replaced at this size read 0.64 and was 17 %% slower per "
                           "iteration" % (avg1 * 1e3, lay1 / 1e6))
    if frac > 1.0:      # physical bytes faster than HBM can deliver them: the working set is served by a cache
        roof["bound"] = "cache"
        roof["frac_exceeds_hbm_peak"] = True
    # SURVEY 8d's own figure: ALGORITHMIC bytes (12 B per nonzero, row pointers, x once, y twice) / the same time /
    # peak.  Above 1 the layout moves fewer bytes than that count and the product is not an HBM stream of it.
    roof["frac_survey8d"] = alg1 / (avg1 * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["bound_survey8d"] = "cache" if roof["frac_survey8d"] > 1.0 else "hbm"
    if info["xlds"] == 3:
        roof["csb_lockstep"] = s.get_option("csb_lockstep_mode1")   # chunks per wave and lock-step step (0: free-running sweep)
    if traffic:
        roof["traffic"] = traffic.get("bytes_per_launch")
        roof["traffic_detail"] = traffic
    kernels = {
        "spmv_mode2": {"avg_launch_us": avg2 * 1e3, "bytes_per_launch": lay2, "gbps": lay2 / (avg2 * 1e-3) / 1e9,
                       "frac": lay2 / (avg2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "effective_gbps": alg2 / (avg2 * 1e-3) / 1e9},
        "update_xw": {"avg_launch_us": avg3 * 1e3, "bytes_per_launch": 40 * n,
                      "gbps": 40 * n / (avg3 * 1e-3) / 1e9, "frac": 40 * n / (avg3 * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    return roof, kernels, (alg1, alg2, lay1, lay2)


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
    """Only the dominant kernel: 3 warm + `reps` back-to-back launches of the mode-1 product.  Under
    `rocprofv3 --kernel-trace --stats` the kernel's average duration x kernel_launches_per_product is
    roofline.avg_launch_us of the same line."""
    spec = HEADLINE if args.workload == "auto" else args.workload
    s, d_b, facts, _ = build_workload(spec, None)
    info = s.info()
    reps = 200 if facts["nnz"] < 50_000_000 else (50 if facts["nnz"] < 200_000_000 else 10)
    avg1 = s.bench_kernel(1, reps)
    lay1 = info["csr_bytes"] + 8 * facts["n"] + 16 * facts["m"]
    lpp = s.get_option("launches_mode1")
    ach = lay1 / (avg1 * 1e-3) / 1e9
    print(json.dumps({"workload": spec, "env": {k: v for k, v in os.environ.items() if k.startswith("LSQRHIP_")},
                      "roofline": {"kernel": describe_layout(info)[1] + " (aprod mode 1)", "bytes_per_launch": lay1,
                                   "avg_launch_us": avg1 * 1e3, "kernel_launches_per_product": lpp,
                                   "avg_kernel_launch_us": avg1 * 1e3 / lpp, "launches": reps, "achieved": ach,
                                   "frac": ach / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s"}}), flush=True)


def side_workload(spec, env, note, K, traffic):
    """An additional instance of the same path (HBM-resident Poisson, configs[3]): a short solve and
    the product's roofline.  Never fails the headline measurement."""
    from lsqr_amd import capi
    try:
        t_build = time.perf_counter()
        s, d_b, facts, _ = build_workload(spec, env, itnlim=K)
        t_build = time.perf_counter() - t_build
        d_x = capi.DeviceBuffer(8 * max(facts["n"], 1))
        s.atol = s.btol = s.conlim = 0.0
        s.set_option("graph_iters", min(K + (K & 1), 50))
        s.itnlim = min(4, K)
        s.solve_device(d_b.ptr.value, d_x.ptr.value, facts["damp"])
        dt, r, restarts = timed_solve(s, d_b, d_x, facts["damp"], K)
        reps = 100 if facts["nnz"] < 200_000_000 else 10
        roof, kernels, (alg1, alg2, lay1, lay2) = product_roofline(s, facts, reps, traffic)
        out = {"workload": f"{spec} m={facts['m']} n={facts['n']} nnz={facts['nnz']} damp={facts['damp']}",
               "variant": note, "env": env or {}, "n_gpus": 1, "steps": K, "value": K / dt, "unit": "it/s",
               "ms_per_step": 1e3 * dt / K, "restarts": restarts, "generate_and_build_s": t_build,
               "iter_bytes_layout": lay1 + lay2 + 40 * facts["n"],
               "iter_gbps_layout": (lay1 + lay2 + 40 * facts["n"]) * K / dt / 1e9,
               "roofline": roof, "kernels": kernels}
        del s, d_b, d_x
        return out
    except Exception as e:      # noqa: BLE001
        return {"workload": spec, "variant": note, "error": repr(e)}


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


def live_traffic(plan, timeout=600):
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
    want_n1 = extras and not args.no_scaling_ref
    from lsqr_amd.dist_bench import DEFAULT_SPEC

    # PMC passes first, as children, before this process touches the GPU
    plan = [[spec, {}]]
    if extras:
        plan += [[sp, env] for sp, env, _ in HBM_INSTANCES]
    if want_n1:
        plan += [[DEFAULT_SPEC, {}]]
    traffic, traffic_note = ({}, "--traffic off")
    if args.traffic == "live" and not args.no_roofline:
        traffic, traffic_note = live_traffic(plan)

    def tr(sp, env):
        return traffic.get((sp, json.dumps(env or {}, sort_keys=True), ()))

    import torch
    from lsqr_amd import capi
    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible; the HIP path has no CPU fallback")
    K, W = args.steps, args.warmup
    s, d_b, facts, host = build_workload(spec, None, itnlim=K)
    d_x = capi.DeviceBuffer(8 * max(facts["n"], 1))
    # graph batch: a divisor of K when there is a good one (no predicated-off tail iterations in
    # the timed solve), else up to 50 iterations (launches past the stop are ~us-scale no-ops)
    gi = next((g for g in (100, 50, 40, 32, 26, 20, 16) if K % g == 0), min(50, K + (K & 1)))
    s.set_option("graph_iters", gi)
    s.atol = s.btol = s.conlim = 0.0
    # setup, untimed: the graphs of this batch size are captured and instantiated by the first solve, and the
    # interpreter and the device clocks settle over a few more (a 20-iteration solve is 0.6 ms of device work:
    # the first timed call after an idle start otherwise measures the ramp, +6 %)
    # ... a 20-iteration solve still speeds up by 4 % over its first ~30 repeats, profiles/r03/perf_misc.txt): at
    # least 3 solves and at least 60 ms of them
    t_warm, n_warm = time.perf_counter(), 0
    while n_warm < 3 or (time.perf_counter() - t_warm < 0.06 and n_warm < 400):
        timed_solve(s, d_b, d_x, facts["damp"], min(K, 100))
        n_warm += 1
    if W > 0:                                   # the W warm-up steps of the contract, through the timed path
        timed_solve(s, d_b, d_x, facts["damp"], W)
    dt, r, restarts = timed_solve(s, d_b, d_x, facts["damp"], K)

    cfgname = " (BASELINE.json configs[1])" if spec == HEADLINE else ""
    out = {
        "metric": "lsqr_iterations_per_sec", "value": K / dt, "unit": "it/s", "n_gpus": 1,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
        "scaling": SCALING_N1 if spec == HEADLINE else "n/a (one GPU, one workload)",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{facts['name']} m={facts['m']} n={facts['n']} nnz={facts['nnz']} "
                               f"damp={facts['damp']}{cfgname}",
                   "graph_iters": gi, "restarts": restarts},
        "result": {"istop": r.istop, "itn": r.itn, "anorm": r.anorm, "rnorm": r.rnorm},
    }
    if not args.no_roofline:
        reps = min(max(K, 500), 1000) if facts["nnz"] < 50_000_000 else (100 if facts["nnz"] < 200_000_000 else 10)
        roof, kernels, (alg1, alg2, lay1, lay2) = product_roofline(s, facts, reps, tr(spec, {}))
        if roof["traffic"] is None:
            roof["traffic_note"] = traffic_note
        out["roofline"] = roof
        out["kernels"] = kernels
        vec = 40 * facts["n"]
        out["iter_bytes_layout"] = lay1 + lay2 + vec
        out["iter_gbps_layout"] = (lay1 + lay2 + vec) * K / dt / 1e9
        out["iter_bytes_survey8d"] = alg1 + alg2 + vec
        out["iter_gbps_survey8d_effective"] = (alg1 + alg2 + vec) * K / dt / 1e9
    del s, d_x
    if extras and not args.no_roofline:
        out["roofline_hbm"] = [side_workload(sp, env, note, 50, tr(sp, env)) for sp, env, note in HBM_INSTANCES]
        # the general short-row kernel (no value dictionary, no repeating rows needed) on an HBM-resident instance,
        # at top level: what a matrix without config 2's special structure gets from this library
        g_inst = out["roofline_hbm"][GENERAL_INSTANCE]
        if "roofline" in g_inst:
            gen = dict(g_inst["roofline"])
            gen["workload"] = g_inst["workload"]
            gen["rocprof_summary"] = "profiles/r05/poisson4000_val8_roofline.txt"
            out["roofline_general"] = gen
        else:
            out["roofline_general"] = g_inst
    if want_n1:
        n1 = side_workload(DEFAULT_SPEC, {}, "BASELINE.json configs[3], whole on one GPU", 40, tr(DEFAULT_SPEC, {}))
        n1["note"] = "the --gpus N > 1 lines shard this matrix by row blocks: compare their value with this one"
        out["strong_scaling_n1"] = n1
    if extras and not args.no_roofline and args.configs == "on":
        # every other BASELINE configuration a single GPU holds, measured by this run (no PMC pass: each would build
        # the matrix twice more; `traffic` of the column-swept kernel is on strong_scaling_n1)
        out["roofline_configs"] = [side_workload(sp, {}, note, k, None) for sp, k, note in ROOFLINE_CONFIGS]
    if args.cpu_iters > 0 and host is not None:
        out["cpu_baseline"] = cpu_baseline(host, args.cpu_iters)
    elif args.cpu_iters > 0:
        out["cpu_baseline"] = None   # workload generated in HBM only; the CPU sample is quoted on config 2
    print(json.dumps(out), flush=True)


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
