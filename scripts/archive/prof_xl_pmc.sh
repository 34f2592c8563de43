#!/bin/bash
# PMC look at the LDS-panel kernel on config 3 literal (counters in passes of their own).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/xl_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in LDSBankConflict LdsUtil MemUnitStalled LdsLatency; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -o p -- python3 "$R/scripts/big_off64.py" random:4000000:1000000:1000 > "$OUT/$C.out" 2> "$OUT/$C.err"
  F=$(find "$OUT/$C" -name "*counter_collection.csv" | head -1)
  python3 - "$F" $C <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "xlw" in r["Kernel_Name"] or "panel_combine" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{sys.argv[2]:18s} {k:42s} n={len(v):3d} mean={sum(v)/len(v):.3f}")
PY
done
