#!/usr/bin/env python3
"""Timeline of ONE short solve (K = 20 at config 2, the shape of the driver's round-end bench run):
run under  rocprofv3 --kernel-trace --hip-trace --output-format csv -d DIR -o t -- python3 scripts/short_solve_timeline.py
then  python3 scripts/short_solve_timeline.py --parse DIR  prints the kernels and HIP calls of the last solve."""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    from lsqr_amd import capi, problems as P
    from lsqr_amd.solver import lsqr_solver_ez
    import torch
    K = int(os.environ.get("K", "20"))
    p = P.poisson2d(1000, 1000)
    s = lsqr_solver_ez().initialize(p.m, p.n, p.a, p.irow, p.icol)
    d_b = capi.DeviceBuffer.from_array(p.b)
    d_x = capi.DeviceBuffer(8 * p.n)
    s.set_option("graph_iters", min(K, 100))
    s.itnlim = K
    for _ in range(3):
        s.solve_device(d_b.ptr.value, d_x.ptr.value, 0.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve_device(d_b.ptr.value, d_x.ptr.value, 0.0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    print("wall us per solve:", " ".join(f"{t:.1f}" for t in ts), " => it/s", K / (min(ts) * 1e-6), flush=True)


def parse(d):
    kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    hf = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
    ks = sorted(csv.DictReader(open(kf)), key=lambda r: int(r["Start_Timestamp"]))
    # the last solve: from the last k_sumsq3 on
    idx = max(i for i, r in enumerate(ks) if "k_start" in r["Kernel_Name"] or "k_sumsq3" in r["Kernel_Name"])
    ks = ks[idx:]
    t0 = int(ks[0]["Start_Timestamp"])
    prev_end = t0
    print("kernels of the last solve (start us, dur us, gap before us):")
    for r in ks:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {(st - prev_end) / 1e3:8.1f}  {r['Kernel_Name'][:70]}")
        prev_end = en
    print(f"device span of the solve: {(prev_end - t0) / 1e3:.1f} us")
    if hf:
        hs = sorted(csv.DictReader(open(hf[0])), key=lambda r: int(r["Start_Timestamp"]))
        # host calls in the window [t0 - 200 us, last kernel end + 200 us]
        print("HIP calls around it (start us rel. first kernel, dur us):")
        for r in hs:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if t0 - 300_000 <= st <= prev_end + 300_000:
                print(f"  {(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  {r['Function']}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
