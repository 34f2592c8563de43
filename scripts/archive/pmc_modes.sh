#!/bin/bash
# PMC comparison of the mode-1 and mode-2 products of one workload (counters in passes of their own).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SPEC=${1:-random:10000000:10000000:100}
OUT=$R/gpurun_out/pmc_modes
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TA_BUSY_avr" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCP_PENDING_STALL_CYCLES_sum" "LDSBankConflict SQ_LDS_BANK_CONFLICT"; do
  T=$(echo $C | tr ' ' '_')
  rm -rf "$OUT/$T"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/$T" -o p -- python3 "$R/scripts/order_ab.py" $SPEC > "$OUT/$T.out" 2> "$OUT/$T.err"
  F=$(find "$OUT/$T" -name "*counter_collection.csv" | head -1)
  [ -z "$F" ] && { echo "$T: no counters"; tail -2 "$OUT/$T.err"; continue; }
  python3 - "$F" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_spmv_csb" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
# order_ab.py: (2,1),(1,2),(2,1) x (3 warm + 10) products x 3 launches: label by position
acc = collections.defaultdict(list)
per = 13 * 3
ctrs = sorted(set(r["Counter_Name"] for r in rows))
for c in ctrs:
    rc = [r for r in rows if r["Counter_Name"] == c]
    seq = [2, 1, 1, 2, 2, 1]
    for k, r in enumerate(rc):
        blk = k // per
        if blk < len(seq):
            acc[(c, seq[blk])].append(float(r["Counter_Value"]))
for (c, m), v in sorted(acc.items()):
    print(f"{c:34s} mode {m}: n={len(v):4d} mean per launch = {sum(v)/len(v):.5g}")
PY
done
