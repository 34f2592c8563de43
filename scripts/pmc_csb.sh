#!/bin/bash
# PMC counters of the column-swept products (csb.h) of one workload, per PRODUCT (all its launches summed),
# mode 1 and mode 2; every counter set in a pass of its own.  usage: pmc_csb.sh SPEC [TAG]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SPEC=${1:-random:10000000:10000000:100}
TAG=${2:-pmc_csb}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== $SPEC [$(env | grep ^LSQRHIP_ | tr '\n' ' ')]"
SETS=${PMC_SETS:-"TCC_HIT_sum,TCC_MISS_sum TCP_TCC_READ_REQ_sum,TCP_TOTAL_CACHE_ACCESSES_sum FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE,TA_BUSY_avr"}
for CS in $SETS; do
  C=$(echo $CS | tr "," " ")
  T=$(echo $C | tr ' ' '_')
  rm -rf "$OUT/$T"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$T" -o p -- python3 "$R/scripts/pmc_products.py" $SPEC > "$OUT/$T.out" 2> "$OUT/$T.err"
  F=$(find "$OUT/$T" -name "*counter_collection.csv" | head -1)
  [ -z "$F" ] && { echo "$T: no counters"; tail -2 "$OUT/$T.err"; continue; }
  python3 - "$F" "$OUT/$T.out" <<'PY'
import csv, sys, collections
lay = [l.split() for l in open(sys.argv[2]) if l.startswith("LAYOUT")][0]
times = [l.split() for l in open(sys.argv[2]) if l.startswith("TIMES_MS")]
L1, L2 = int(lay[3]), int(lay[4])
S1, S2 = int(lay[9]), int(lay[10])
# dispatches = [k_csb_xmax, unless the products keep the piece maxima themselves] + sweeps + [k_csb_combine if splits]
XM = 0 if lay[-1] == "xfold" else 1
L1 -= XM + (1 if S1 > 1 else 0)
L2 -= XM + (1 if S2 > 1 else 0)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_spmv_csb" in r["Kernel_Name"]]
for c in sorted(set(r["Counter_Name"] for r in rows)):
    rc = sorted((r for r in rows if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))
    v = [float(r["Counter_Value"]) for r in rc]
    m1 = v[3 * L1:13 * L1]
    m2 = v[13 * L1 + 3 * L2:13 * L1 + 13 * L2]
    print(f"{c:30s} per product: mode 1 {sum(m1) / 10:.5g} ({L1} launches)   mode 2 {sum(m2) / 10:.5g} ({L2} launches)   [n={len(v)}]")
if times:
    print("   ", " ".join(lay), "|", " ".join(times[0]))
PY
done
