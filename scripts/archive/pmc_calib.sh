#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_calib; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_HIT_sum"; do
  T=$(echo $C | tr ' ' '_'); rm -rf $OUT/$T
  rocprofv3 --pmc $C --output-format csv -d $OUT/$T -o p -- python3 $R/scripts/pmc_calib.py > $OUT/$T.out 2> $OUT/$T.err
  F=$(find $OUT/$T -name "*counter_collection.csv" | head -1)
  [ -z "$F" ] && { echo "$T: none"; tail -2 $OUT/$T.err; continue; }
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if any(t in k for t in ("k_scale", "k_copy(", "k_dot")) and int(r["Grid_Size"]) > 1000:
        acc[(k[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = v[-3:]
    print(f"{c:28s} {k:42s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
done
