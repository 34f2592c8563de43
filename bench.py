#!/usr/bin/env python3
"""bench.py -- LSQR iterations/s + aprod SpMV GB/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE LSQR iteration (mode-1 SpMV + mode-2 SpMV + x/w update + the scalar
recurrences) over the whole system.  The timed region is one `solve` of exactly K
iterations (atol = btol = conlim = 0, itnlim = K -> istop = 5) with the matrix, b and x
already resident in HBM; W warm-up iterations run first as a separate solve.

N = 1 workload: BASELINE.json configs[1] -- 1M x 1M 5-point Poisson (nnz 4 996 000),
damp = 0.  N > 1: the row-block sharded solve (lsqr_amd/dist.py).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = mode-1 SpMV, live HIP event
timing, bytes of the layout the build chose) and `cpu_baseline` (the reference's own CPU path,
1 core, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed LSQR iterations (default 2000; 400 for --gpus > 1)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed iterations first (default steps / 10)")
    ap.add_argument("--workload", default="auto",
                    help="auto | poisson2d:NX:NY | random:M:N:PER_ROW | powerlaw:M:N:DMAX")
    ap.add_argument("--cpu-iters", type=int, default=1000, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-scaling-ref", action="store_true",
                    help="skip the N = 1 point of the multi-GPU series (configs[3] whole on this GPU)")
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


def scaling_series_n1(torch):
    """N = 1 of the series `--gpus 2/4/8` continues: BASELINE configs[3] (10M x 10M, 100 per row,
    10^9 nonzeros, damp 1e-3 -- lsqr_amd/dist_bench.py) whole on this GPU.  The N > 1 lines are
    strong scaling of THIS workload, not of configs[1] above; outside the timed region."""
    from lsqr_amd import capi, devgen
    from lsqr_amd.dist_bench import DEFAULT_SPEC
    try:
        full = devgen.generate(DEFAULT_SPEC)
        cfg = devgen.parse_spec(DEFAULT_SPEC)
        d_x = capi.DeviceBuffer(8 * cfg["n"])
        full.solver.atol = full.solver.btol = full.solver.conlim = 0.0
        full.solver.itnlim = 4
        full.solver.solve_device(full.d_b.ptr.value, d_x.ptr.value, cfg["damp"])
        kr = 40
        full.solver.itnlim = kr
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = full.solver.solve_device(full.d_b.ptr.value, d_x.ptr.value, cfg["damp"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"workload": f"{DEFAULT_SPEC} nnz={full.nnz} damp={cfg['damp']} (BASELINE.json configs[3], whole on one GPU)",
                "n_gpus": 1, "steps": r.itn, "value": r.itn / dt, "unit": "it/s", "ms_per_step": 1e3 * dt / r.itn,
                "note": "the --gpus N > 1 lines shard this matrix by row blocks: compare their value with this one"}
    except Exception as e:      # never fail the headline measurement over the side one
        return {"error": repr(e)}


def run_single(args):
    import torch
    from lsqr_amd import capi
    from lsqr_amd.solver import lsqr_solver_ez

    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible; the HIP path has no CPU fallback")
    from lsqr_amd import devgen
    spec = "poisson2d:1000:1000" if args.workload == "auto" else args.workload
    K, W = args.steps, args.warmup
    cfg = devgen.parse_spec(spec)
    nnz_est = cfg["m"] * (5 if cfg["kind"] == "poisson2d" else cfg.get("per_row", 30))
    host_ok = nnz_est <= 60_000_000          # the host copy only exists for the CPU baseline
    if host_ok:
        p = make_problem(spec)
        s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol, itnlim=K)
        d_b = capi.DeviceBuffer.from_array(p.b)
    else:                                     # generated in HBM (bit-identical generator, csrc/gen_api.h)
        dp = devgen.generate(spec, itnlim=K)
        s, d_b = dp.solver, dp.d_b

        class _P:                             # what the report below needs
            name, m, n, nnz, damp = spec, dp.m, dp.n, dp.nnz, dp.damp
        p = _P
    d_x = capi.DeviceBuffer(8 * max(p.n, 1))
    # graph batch: a divisor of K when there is a good one (no predicated-off tail iterations in
    # the timed solve), else up to 50 iterations (launches past the stop are ~us-scale no-ops)
    gi = next((g for g in (100, 50, 40, 32, 26, 20, 16) if K % g == 0), min(50, K + (K & 1)))
    s.set_option("graph_iters", gi)

    if W > 0:
        s.itnlim = W
        s.solve_device(d_b.ptr.value, d_x.ptr.value, p.damp)
    # EXACTLY K iterations.  With atol = btol = conlim = 0 the default workloads never stop before
    # the limit (1000^2 Poisson needs > 10^5 iterations), but a well-conditioned --workload can reach
    # machine precision first (random 4M x 1M, damp 1e-3: 50 iterations): the solve is then started
    # again on the same b until K iterations have run; `restarts` in the report counts that.
    done, restarts, loop_ms = 0, 0, 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while done < K:
        s.itnlim = K - done
        r = s.solve_device(d_b.ptr.value, d_x.ptr.value, p.damp)
        done += r.itn
        loop_ms += s.last_timing().loop_ms
        if done < K:
            restarts += 1
            if r.itn == 0:
                raise SystemExit(f"bench.py: workload {spec} stops at iteration 0 (b = 0?)")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert done == K and (restarts > 0 or r.istop == 5), (done, r.itn, r.istop)
    tm = s.last_timing()
    tm.loop_ms = loop_ms

    out = {
        "metric": "lsqr_iterations_per_sec", "value": K / dt, "unit": "it/s", "n_gpus": 1,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{p.name} m={p.m} n={p.n} nnz={p.nnz} damp={p.damp} "
                               f"(BASELINE.json configs[1])" if spec == "poisson2d:1000:1000" else
                               f"{p.name} m={p.m} n={p.n} nnz={p.nnz} damp={p.damp}",
                   "graph_iters": gi, "device_loop_ms": tm.loop_ms, "restarts": restarts},
        "result": {"istop": r.istop, "itn": r.itn, "anorm": r.anorm, "rnorm": r.rnorm},
        "iter_bytes": tm.spmv1_bytes + tm.spmv2_bytes + tm.vec_bytes,
        "iter_gbps": (tm.spmv1_bytes + tm.spmv2_bytes + tm.vec_bytes) * K / dt / 1e9,
    }

    if not args.no_roofline:
        # Same K iterations again, eager launches with HIP events around each hot kernel
        # (recorded on the stream the kernels run on).
        s.set_option("time_kernels", 1)
        s.itnlim = r.itn                      # the last (or only) solve of the timed region again
        r2 = s.solve_device(d_b.ptr.value, d_x.ptr.value, p.damp)
        t2 = s.last_timing()
        s.set_option("time_kernels", 0)
        assert r2.itn == r.itn and r2.anorm == r.anorm
        in_loop = [t2.spmv1_ms / max(t2.spmv1_launches, 1), t2.spmv2_ms / max(t2.spmv2_launches, 1),
                   t2.update_ms / max(t2.update_launches, 1)]
        # K back-to-back launches of each hot kernel inside ONE event pair: the per-launch
        # average rocprofv3's kernel trace reports (kernels abut on the stream; a start/stop
        # event pair per launch adds ~2 us of marker latency to a 17 us kernel).
        reps = min(K, 1000) if p.nnz < 50_000_000 else max(10, min(K, 40))
        avg1, avg2, avg3 = (s.bench_kernel(w, reps) for w in (1, 2, 3))
        # Bytes one product must move IN THE LAYOUT THE BUILD CHOSE: the matrix as
        # stored (lsqrhip_info: sliced-ELL / row windows, 1- or 8-byte values, 2- or 4-byte
        # columns) + x once + y read and written.  SURVEY 8(d)'s CSR figure (8-byte values, 4-byte
        # columns) is reported beside it; with a value dictionary it exceeds what is moved.
        info = s.info()
        fmt1 = info["csr_bytes"] + 8 * p.n + 16 * p.m
        fmt2 = info["csrt_bytes"] + 8 * p.m + 16 * p.n
        if info["sell"] == 2:
            layout, kname = "sell, packed 16-byte records", "k_spmv_sellp"
        elif info["sell"]:
            layout, kname = "sell", "k_spmv_sell"
        elif info["xlds"] == 2:
            layout, kname = "lds-panels (wave windows)", "k_spmv_xlw + k_panel_combine"
        elif info["panels"] > 1:
            layout, kname = "l2-panels", "k_spmv_fused + k_panel_combine"
        else:
            layout, kname = "row-windows", "k_spmv_fused"
        # roofline.achieved follows the contract: ALGORITHMIC bytes of the product (fp64
        # values, int32 columns, row pointers, x once, y read + written) / average launch time.
        # The bytes the chosen layout really moves are reported beside it (layout_*): with the
        # value dictionary / 16-bit columns / sliced ELL they are fewer, so `achieved` can exceed
        # what the memory system delivered -- `traffic` (PMC counters) is the physical figure.
        ach = t2.spmv1_bytes / (avg1 * 1e-3) / 1e9
        lay = fmt1 / (avg1 * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(spec, {}).get("spmv_mode1_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        compressed = fmt1 < 0.98 * t2.spmv1_bytes
        out["roofline"] = {"bound": "hbm", "kernel": f"{kname} (aprod mode 1)",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "bytes_per_launch": t2.spmv1_bytes,
                           "avg_launch_us": avg1 * 1e3, "launches": reps,
                           "in_loop_event_pair_us": in_loop[0] * 1e3,
                           "format": {"layout": layout, "value_bytes": info["value_bytes"],
                                      "col_bytes": info["col_bytes"], "dict_entries": info["dict_entries"]},
                           "layout_bytes_per_launch": fmt1, "layout_gbps": lay, "layout_frac": lay / HBM_PEAK_GBS,
                           "note": ("algorithmic bytes = SURVEY 8d CSR count (12 B per nonzero); the layout in use "
                                    "stores %d B per nonzero, so achieved is an EFFECTIVE rate -- layout_gbps and "
                                    "traffic are the physical ones" % (info["value_bytes"] + info["col_bytes"]))
                                   if compressed else "layout moves the algorithmic bytes"}
        out["iter_format_bytes"] = fmt1 + fmt2 + t2.vec_bytes
        out["iter_format_gbps"] = (fmt1 + fmt2 + t2.vec_bytes) * K / dt / 1e9
        out["kernels"] = {
            "spmv_mode2": {"avg_launch_us": avg2 * 1e3, "bytes_per_launch": t2.spmv2_bytes,
                           "gbps": t2.spmv2_bytes / (avg2 * 1e-3) / 1e9, "layout_bytes_per_launch": fmt2,
                           "layout_gbps": fmt2 / (avg2 * 1e-3) / 1e9, "in_loop_event_pair_us": in_loop[1] * 1e3},
            "update_xw": {"avg_launch_us": avg3 * 1e3, "bytes_per_launch": t2.vec_bytes,
                          "gbps": t2.vec_bytes / (avg3 * 1e-3) / 1e9, "in_loop_event_pair_us": in_loop[2] * 1e3},
        }

    if spec == "poisson2d:1000:1000" and not args.no_scaling_ref:
        out["strong_scaling_n1"] = scaling_series_n1(torch)
    if args.cpu_iters > 0 and host_ok:
        out["cpu_baseline"] = cpu_baseline(p, args.cpu_iters)
    elif args.cpu_iters > 0:
        out["cpu_baseline"] = None   # workload generated in HBM only; the CPU sample is quoted on config 2
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("LSQR_BENCH_FORCE_DIST") == "1":
        from lsqr_amd.dist_bench import run_distributed
        run_distributed(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
