#!/bin/bash
# PMC look at the L2-panel row-window kernel on the N=8 shard shape (counters in passes of their own).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SPEC=${1:-random:1250000:10000000:100}
OUT=$R/gpurun_out/panel_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "GRBM_GUI_ACTIVE TA_BUSY_avr" "MemUnitStalled MeanOccupancyPerCU" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "LDSBankConflict" "VALUBusy SALUBusy" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$T" -o p -- python3 "$R/scripts/kernel_times.py" $SPEC 10 > "$OUT/$T.out" 2> "$OUT/$T.err"
  F=$(find "$OUT/$T" -name "*counter_collection.csv" | head -1)
  [ -z "$F" ] && { echo "$T: no counters (see $OUT/$T.err)"; tail -2 "$OUT/$T.err"; continue; }
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_spmv_fused" in k or "panel_combine" in k or "sellp" in k or "xlw" in k:
        acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{c:32s} {k:62s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
done
