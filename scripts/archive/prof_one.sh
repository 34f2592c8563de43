#!/bin/bash
# kernel durations of one workload's products under rocprofv3 (kernel trace + stats).  usage: prof_one.sh TAG SPEC
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; SPEC=$2
OUT=$R/gpurun_out/prof_one/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$R/scripts/kernel_times.py" $SPEC 10 > "$OUT/out.txt" 2> "$OUT/err.txt"
grep -v amdgpu "$OUT/out.txt"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_spmv" in r["Name"] or "combine" in r["Name"]:
        print("%-60s calls=%5s avg_ns=%12s min=%s max=%s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r.get("MinNs"), r.get("MaxNs")))
PY
