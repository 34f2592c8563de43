"""The shape of the ONE JSON line bench.py prints (CPU test, no GPU): the contract's keys, the `roofline` and
`cpu_baseline` objects -- for both line shapes, the single-GPU one and the distributed one -- held on the lines
committed under profiles/ (what the driver's BENCH / SCALE files are made of) and on the key tuples the
distributed leg builds its line from (lsqr_amd/dist_bench.py)."""
import glob
import json
import os

from lsqr_amd import dist_bench

ROOT = os.path.join(os.path.dirname(__file__), "..")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def last_round_dir():
    """the latest profiles/rNN that holds the lines (a new round's directory starts empty)"""
    dirs = [d for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]")))
            if all(os.path.exists(os.path.join(d, n)) for n in ("bench_default.json", "bench_default_k20.json",
                                                                "engine_1rank_shard8.json"))]
    assert dirs, "no profiles/rNN with the committed bench lines"
    return dirs[-1]


def load(name):
    with open(os.path.join(last_round_dir(), name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def check_line(d, n_gpus_min=1, cpu=True):
    for k in CONTRACT:
        assert k in d, k
    assert d["metric"] == "lsqr_iterations_per_sec" and d["unit"] == "it/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] in ("f64", "f32")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] / 1e3 - 1.0) < 1e-6 * max(1, d["n_gpus"]) + 1e-3
    for k in dist_bench.ROOFLINE_KEYS:
        assert k in d["roofline"], k
    r = d["roofline"]
    assert r["bound"] in ("hbm", "cache") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if not cpu:      # (a run made with --cpu-iters 0: the key is there, the leg was not run)
        assert "cpu_baseline" in d
        return
    for k in dist_bench.CPU_BASELINE_KEYS:
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] >= 1


def test_distributed_leg_names_the_contract_keys():
    for k in CONTRACT:
        assert k in dist_bench.LINE_KEYS, k
    assert "roofline" in dist_bench.LINE_KEYS and "cpu_baseline" in dist_bench.LINE_KEYS
    assert set(dist_bench.ROOFLINE_KEYS) == {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(dist_bench.CPU_BASELINE_KEYS) == {"value", "unit", "cores", "kind", "sample"}


def round_of_lines():
    return int(os.path.basename(last_round_dir())[1:])


def check_survey8d(r):
    """round 4 on: SURVEY 8d's own fraction (algorithmic bytes / time / peak) beside the physical one"""
    assert abs(r["frac_survey8d"] - r["effective_gbps"] / r["peak"]) < 1e-9
    assert r["bound_survey8d"] == ("cache" if r["frac_survey8d"] > 1.0 else "hbm")


LINE_MAX = 4096


def raw_line(name):
    with open(os.path.join(last_round_dir(), name)) as f:
        return f.read().strip().splitlines()[-1]


def system_of(workload: str):
    """the tokens of config.workload that name the SYSTEM: spec, m, n, nnz, damp (what follows says how it is laid out)"""
    return workload.split(" (")[0].split()


def test_the_line_fits_the_drivers_reader():
    """Round 5's line grew to 22 KB and the driver could not parse it (BENCH_r05.parsed = null).  From round 6 on every
    committed driver-form line is at most 4 KB, and bench.emit drops optional keys rather than exceed it."""
    import io
    import sys

    import bench
    assert bench.LINE_MAX == LINE_MAX
    if round_of_lines() >= 6:
        for name in ("bench_default.json", "bench_default_k20.json", "engine_1rank_shard8.json"):
            assert len(raw_line(name)) <= LINE_MAX, (name, len(raw_line(name)))
    # emit(): a line made too long on purpose loses `notes`, then `variants`, `configs` ... -- never the contract keys
    line = {k: 1 for k in CONTRACT}
    line.update(roofline={k: 0 for k in dist_bench.ROOFLINE_KEYS}, cpu_baseline={k: 0 for k in dist_bench.CPU_BASELINE_KEYS},
                notes=["x" * 3000], configs=[{"workload": "y" * 200} for _ in range(12)], variants={"plain": {"value": 1.0}})
    old, sys.stdout = sys.stdout, io.StringIO()
    try:
        bench.emit(line, {"big": "z" * 10000}, os.path.join(os.environ.get("TMPDIR", "/tmp"), "bench_detail_test.json"))
        text = sys.stdout.getvalue().strip()
    finally:
        sys.stdout = old
    assert len(text) <= LINE_MAX and "\n" not in text
    d = json.loads(text)
    for k in CONTRACT + ("roofline", "cpu_baseline", "detail"):
        assert k in d, k
    assert "notes" in d["dropped_for_length"] and "variants" in d


def check_round6_roofline(r):
    """round 6 on: `achieved` / `frac` ARE SURVEY 8d's figure (algorithmic bytes / time / peak); the layout's own
    bytes are beside it, and the transposed product's fraction"""
    assert "SURVEY 8d" in r["bytes_are"] and r["bytes_per_launch"] > 0
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["frac_layout"] > 0 and r["frac_mode2"] > 0 and r["avg_launch_us_mode2"] > 0


def test_committed_single_gpu_lines():
    if round_of_lines() >= 6:
        for name in ("bench_default.json", "bench_default_k20.json"):
            d = load(name)
            check_line(d)
            assert d["n_gpus"] == 1 and d["scaling"] == "strong"
            # the headline IS the north star's workload: configs[3] whole on one GPU
            assert system_of(d["config"]["workload"])[0] == dist_bench.DEFAULT_SPEC and "configs[3]" in d["config"]["workload"]
            r = d["roofline"]
            check_round6_roofline(r)
            assert r["bound"] == "hbm" and 0.5 < r["frac"] <= 1.0 and "k_spmv_csb" in r["kernel"]
            assert r["traffic"] is not None and 0.9 * r["bytes_per_launch"] < r["traffic"] < 2 * r["bytes_per_launch"]
            assert r["kernel_launches_per_product"] >= 1
            assert "scaled" in d["cpu_baseline"]["value_is"] or d["cpu_baseline"]["sample"]
            # the other BASELINE configurations and the rank blocks, compact
            cf = d["configs"]
            assert [e["workload"] for e in cf] == ["poisson2d:1000:1000", "random:4000000:1000000:1000",
                                                   "powerlaw:5000000:2000000:10000", "random:1250000:10000000:100",
                                                   "random:1250000:10000000:1000"]
            for e in cf:
                assert "error" not in e, e
                assert e["it_s"] > 0 and e["frac_mode1"] > 0 and e["frac_mode2"] > 0
                if e.get("bound", "hbm") == "hbm":
                    assert e["frac_mode1"] <= 1 and e["frac_mode2"] <= 1
            assert cf[3]["engine_world1_ms_per_step"] > 0
            assert os.path.basename(d["detail"]).endswith(".json")
        assert load("bench_default_k20.json")["steps"] == 20 and load("bench_default_k20.json")["warmup"] == 5
        return
    _committed_single_gpu_lines_r05()


def test_n1_and_n_gt_1_lines_name_the_same_system():
    """`value` of the --gpus 1 line and of the --gpus N lines are points of ONE series: the same (m, n, nnz, damp)."""
    import bench
    assert bench.HEADLINE == dist_bench.DEFAULT_SPEC
    if round_of_lines() < 6:
        return
    one = system_of(load("bench_default.json")["config"]["workload"])
    for ov in (0, 1):
        with open(os.path.join(last_round_dir(), f"rccl_shared_gpu_configs3_w8_overlap{ov}.json")) as f:
            d = json.loads(f.read().strip().splitlines()[-1])
        assert system_of(d["config"]["workload"]) == one, (one, d["config"]["workload"])
        assert d["scaling"] == load("bench_default.json")["scaling"] == "strong"


def _committed_single_gpu_lines_r05():
    for name in ("bench_default.json", "bench_default_k20.json"):
        d = load(name)
        check_line(d)
        assert d["n_gpus"] == 1
        assert d["roofline"]["traffic"] is not None          # live PMC passes of the same run
        assert isinstance(d.get("roofline_hbm"), list) and d["roofline_hbm"]
        assert "roofline" in d["strong_scaling_n1"]
        if round_of_lines() >= 4:
            # the N = 1 line measures configs[1]: it says so instead of claiming a scaling mode, and points at the
            # series the --gpus N lines continue
            assert d["scaling"].startswith("n/a") and "strong_scaling_n1" in d["scaling"]
            check_survey8d(d["roofline"])
            for e in d["roofline_hbm"]:
                check_survey8d(e["roofline"])
            n1 = d["strong_scaling_n1"]["roofline"]
            check_survey8d(n1)
            if round_of_lines() == 4:
                # round 4 quoted a builder-measured "ceiling of the access pattern" as a constant; round 5's lock-step
                # sweep runs past it and the keys are gone
                assert n1["ceiling_gbps"] > 0 and abs(n1["of_ceiling"] - n1["achieved"] / n1["ceiling_gbps"]) < 1e-9
            else:
                assert "ceiling_gbps" not in n1 and n1["csb_lockstep"] in (0, 1, 2)
                # every other BASELINE configuration one GPU holds, measured by the same run
                rc = d["roofline_configs"]
                assert [e["workload"].split()[0] for e in rc] == ["random:4000000:1000000:1000", "powerlaw:5000000:2000000:10000",
                                                                  "random:1250000:10000000:1000"]
                for e in rc:
                    assert "error" not in e, e
                    check_survey8d(e["roofline"])
                    for k in dist_bench.ROOFLINE_KEYS:
                        assert k in e["roofline"], k
                    assert 0 < e["roofline"]["frac"] <= 1 and e["value"] > 0
            # the best HBM-resident GENERAL short-row kernel at top level, with the rocprofv3 summary that backs it
            g = d["roofline_general"]
            assert g["bound"] == "hbm" and 0 < g["frac"] <= 1 and "poisson2d:4000:4000" in g["workload"]
            assert os.path.exists(os.path.join(ROOT, g["rocprof_summary"]))
        else:
            assert d["scaling"] == "weak"
    assert load("bench_default_k20.json")["steps"] == 20 and load("bench_default_k20.json")["warmup"] == 5


def test_committed_distributed_line():
    d = load("engine_1rank_shard8.json")        # the N > 1 line shape, forced at world = 1
    check_line(d)
    assert d["scaling"] == "strong" and d["config"]["engine"] in ("c++", "python") and "engine_note" in d["config"]
    if round_of_lines() >= 6:
        check_round6_roofline(d["roofline"])
        assert len(raw_line("engine_1rank_shard8.json")) <= LINE_MAX
    if round_of_lines() >= 4:
        if round_of_lines() < 6:
            check_survey8d(d["roofline"])
        # next to `value`: the same workload on one GPU and the ratio (None at world = 1: nothing to compare), and
        # whether the exchanges ran overlapped
        for k in ("value_1gpu_same_workload", "speedup_vs_1gpu_same_workload", "overlap"):
            assert k in d, k
        assert d["overlap"] in (0, 1)


def test_committed_line_of_configs3_on_eight_rccl_ranks():
    """BASELINE configs[3] as stated (8 ranks, RCCL), run with the ranks sharing one GPU (round 4 on): the N > 1 line in
    full, produced by the C++ engine without a fall-back, carrying the one-handle result of the same iterations and the
    distance to it."""
    if round_of_lines() < 4:
        return
    for ov in (0, 1):
        path = os.path.join(last_round_dir(), f"rccl_shared_gpu_configs3_w8_overlap{ov}.json")
        with open(path) as f:
            d = json.loads(f.read().strip().splitlines()[-1])
        check_line(d, cpu=False)
        assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["overlap"] == ov
        c = d["config"]
        assert c["world_size"] == 8 and c["rows_per_rank"] == [1250000] * 8 and "random:10000000:10000000:100" in c["workload"]
        assert c["engine"] == "c++" and c["engine_note"] is None and c["ranks_share_one_gpu"] is True
        if round_of_lines() >= 6:      # (the one-GPU reference solve itself is in the detail file; its distance is in the line)
            assert len(json.dumps(d)) <= LINE_MAX and d["result"]["itn"] == d["steps"] and d["value_1gpu_same_workload"] > 0
            assert d["sharded_vs_1gpu"]["rnorm_rel"] <= 1e-10 and d["sharded_vs_1gpu"]["anorm_rel"] <= 1e-10
            continue
        ref = d["strong_scaling_ref"]
        assert ref["result"]["itn"] == d["result"]["itn"] == d["steps"]
        assert ref["sharded_vs_1gpu"]["rnorm_rel"] <= 1e-10 and ref["sharded_vs_1gpu"]["anorm_rel"] <= 1e-10
