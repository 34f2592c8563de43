#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a small text summary."""
import csv, glob, os, sys, collections

out = sys.argv[1]

def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("  %-70s calls=%6s total_ns=%12s avg_ns=%10s pct=%6s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

for kind in ("pmc_fetch", "pmc_write"):
    for f in find(kind + "/**/*counter_collection.csv"):
        print("== counters:", os.path.relpath(f, out))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "")[:60], r.get("Counter_Name"))
            agg[k][0] += 1
            agg[k][1] += float(r.get("Counter_Value", 0))
        for (kn, cn), (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
            print("  %-60s %-12s dispatches=%6d  mean=%.1f" % (kn, cn, cnt, tot / max(cnt, 1)))
