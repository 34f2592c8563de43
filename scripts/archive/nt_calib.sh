#!/bin/bash
# PMC tallies of ordinary vs non-temporal loads (scripts/nt_calib.hip) -> gpurun_out/nt_calib.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/nt_calib; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o $OUT/nt_calib $R/scripts/nt_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_HIT_sum"; do
  T=$(echo $C | tr ' ' '_'); rm -rf $OUT/$T
  rocprofv3 --pmc $C --output-format csv -d $OUT/$T -o p -- $OUT/nt_calib > $OUT/$T.out 2> $OUT/$T.err
  F=$(find $OUT/$T -name "*counter_collection.csv" | head -1)
  [ -z "$F" ] && { echo "$T: none"; tail -2 $OUT/$T.err; continue; }
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_read" in k:
        acc[(k[:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{c:28s} {k:26s} n={len(v)} mean={sum(v)/len(v):.6g}   (1.2e9 bytes read per launch)")
PY
done
