#!/usr/bin/env python3
"""Rewrite README.md's results table (between the BENCH_TABLE markers) from the round's committed default bench line.
usage: fill_readme.py profiles/r06"""
import json, os, sys
d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = os.path.basename(d.rstrip("/"))
line = json.loads(open(os.path.join(d, "bench_default.json")).read().strip().splitlines()[-1])
r = line["roofline"]
cf = {e["workload"]: e for e in line["configs"]}


def row(name, layout, e, note=""):
    its = ("%.0f" % e["it_s"]) if e["it_s"] >= 1000 else ("%.1f" % e["it_s"])
    return "| %s | %s | %s | %.0f / %.0f µs | **%.0f %% / %.0f %%**%s |\n" % (
        name, layout, its, e["us_mode1"], e["us_mode2"], 100 * e["frac_mode1"], 100 * e["frac_mode2"], note)


new = ("Measured on one MI355X (round %s; `%s/`; every row is in the ONE line `python bench.py` prints — ≤ 4 KB, the\n"
       "rest in the side file it names).  Fractions are of the 8 TB/s HBM peak on SURVEY §8d's ALGORITHMIC bytes (12 B per\n"
       "nonzero, row pointers, x once, y twice) — the figure the roofline object of the line carries:\n\n"
       "| workload | layout | iterations/s | product, mode 1 / mode 2 | of 8 TB/s, mode 1 / mode 2 |\n|---|---|---|---|---|\n"
       % (int(rnd[1:]), d.rstrip("/")))
new += ("| **10M × 10M random, 100 per row (10⁹ nonzeros): BASELINE configs[3], whole on one GPU — the bench's headline and "
        "the N = 1 point of the `--gpus N` series** | column-swept row blocks, lock-step sweep | **%.1f** (reference on one host "
        "core, scaled from a 2·10⁷-nonzero sample: %.2f) | %.0f / %.0f µs | **%.0f %% / %.0f %%**; PMC traffic %.2f × the "
        "algorithmic bytes |\n" % (line["value"], line["cpu_baseline"]["value"], r["avg_launch_us"], r["avg_launch_us_mode2"],
                                  100 * r["frac"], 100 * r["frac_mode2"], r["traffic"] / r["bytes_per_launch"]))
e = cf["random:1250000:10000000:100"]
new += row("one rank's block of that matrix at N = 8 (1.25M × 10M)",
           "column-swept row blocks, 4 column splits closed by their last arriver", e,
           "; through the sharded engine at world 1: %.3f ms per iteration" % e["engine_world1_ms_per_step"])
new += row("… at 1000 per row (1.25·10⁹ nonzeros)", "column-swept row blocks", cf["random:1250000:10000000:1000"])
new += row("4M × 1M random, 1000 per row (4·10⁹ nonzeros): configs[2] at its literal size", "column-swept row blocks",
           cf["random:4000000:1000000:1000"])
new += row("power-law 5M × 2M, rows up to 10⁴: configs[4]", "column-swept row blocks (Aᵀ: 128 blocks × 2 splits)",
           cf["powerlaw:5000000:2000000:10000"])
e = cf["poisson2d:1000:1000"]
new += ("| 1M × 1M 5-point Poisson: configs[1], K = 20 | row patterns: 1 byte per row, paired rows | %.0f | %.1f / %.1f µs | "
        "lives in the Infinity Cache (42 MB): %.1f × peak on §8d's bytes, reported as `bound: \"cache\"`, no roofline claim; on "
        "a 16M-row instance the same kernel streams at 0.67 of peak on the bytes it moves |\n\n"
        % (e["it_s"], e["us_mode1"], e["us_mode2"], e["frac_mode1"]))
p = os.path.join(root, "README.md")
s = open(p).read()
a = s.index("<!-- BENCH_TABLE_BEGIN")
a = s.index("\n", a) + 1
b = s.index("<!-- BENCH_TABLE_END -->")
open(p, "w").write(s[:a] + new + s[b:])
