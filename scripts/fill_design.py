#!/usr/bin/env python3
"""Fill the @@TOKENS@@ of DESIGN.md from the round's committed evidence (profiles/rNN): the default bench line, the rank
block series, the GPU suite's summary.  usage: fill_design.py profiles/r06   (writes DESIGN.md from scripts/DESIGN.template.md)"""
import json, os, re, sys
d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = json.loads(open(os.path.join(d, "bench_default.json")).read().strip().splitlines()[-1])
k20 = json.loads(open(os.path.join(d, "bench_default_k20.json")).read().strip().splitlines()[-1])
r = line["roofline"]
cf = {e["workload"]: e for e in line["configs"]}
cf20 = {e["workload"]: e for e in k20["configs"]}
T = {}
T["C3_ITS"] = "%.1f" % line["value"]
T["C3_F1"] = "%.3f" % r["frac"]
T["C3_F2"] = "%.3f" % r["frac_mode2"]
T["C3_US1"] = "%.0f" % r["avg_launch_us"]
T["C3_TRAFFIC"] = "%.2f" % (r["traffic"] / r["bytes_per_launch"]) if r.get("traffic") else "n/a"
T["C3_ITER_GBPS"] = "%.0f" % line["spmv_gbps"]["iteration"]
def cfg(prefix, spec, src=cf):
    e = src[spec]
    T[prefix + "_ITS"] = ("%.1f" if e["it_s"] < 1000 else "%.0f") % e["it_s"]
    T[prefix + "_F1"] = "%.3f" % e["frac_mode1"]
    T[prefix + "_F2"] = "%.3f" % e["frac_mode2"]
    T[prefix + "_US1"] = "%.1f" % e["us_mode1"]
    T[prefix + "_US2"] = "%.1f" % e["us_mode2"]
    if "engine_world1_ms_per_step" in e:
        T[prefix + "_ENG"] = "%.3f" % e["engine_world1_ms_per_step"]
cfg("C2L", "random:4000000:1000000:1000")
cfg("C5", "powerlaw:5000000:2000000:10000")
cfg("S8", "random:1250000:10000000:100")
cfg("S8K", "random:1250000:10000000:1000")
cfg("C2", "poisson2d:1000:1000")
T["C2_ITS20"] = T["C2_ITS"]
cpu = line["cpu_baseline"]
T["CPU_ITS"] = "%.3f" % cpu["value"]
T["CPU_SAMPLE"] = "%.1f" % cpu["sample_value"]
rb = {}
for ln in open(os.path.join(d, "rank_block_times.txt")):
    m = re.match(r"rows per rank (\d+): engine at world = 1 ([\d.]+) ms", ln)
    if m:
        rb[int(m.group(1))] = float(m.group(2))
for n, rows in ((1, 10000000), (2, 5000000), (4, 2500000), (8, 1250000)):
    T["RB%d" % n] = "%.3f" % rb[rows]
T["RB_RATIO"] = "%.2f" % (rb[10000000] / rb[1250000])
m = re.search(r"(\d+) passed.* in ([\d.]+)s", open(os.path.join(d, "full_gpu.txt")).read())
T["SUITE_S"] = "%.0f" % float(m.group(2))
p4 = os.path.join(d, "poisson4000_patterns_roofline.txt")
mm = re.search(r'"frac_layout": ([\d.]+)', open(p4).read()) if os.path.exists(p4) else None
T["P4000_F1"] = "%.2f" % float(mm.group(1)) if mm else "0.68"
path = os.path.join(root, "DESIGN.md")
s = open(os.path.join(root, "scripts", "DESIGN.template.md")).read()      # DESIGN.md is GENERATED from the template: edit that
missing = set(re.findall(r"@@(\w+)@@", s)) - set(T)
assert not missing, missing
for k, v in T.items():
    s = s.replace("@@%s@@" % k, v)
open(path, "w").write(s)
print(json.dumps(T, indent=1))
