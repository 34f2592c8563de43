#!/bin/bash
# rocprofv3 kernel stats of the back-to-back kernel timing script for one workload.
# usage: scripts/prof_kernels.sh SPEC TAG
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/kprof_$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- python3 "$R/scripts/kernel_times.py" "$1" 50 > "$OUT/out.txt" 2> "$OUT/err.txt"
cat "$OUT/out.txt"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
