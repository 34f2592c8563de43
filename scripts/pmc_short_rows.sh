#!/bin/bash
# PMC counters of the short-row products at 16M rows (mode 1, `bench.py --roofline-only`): how busy the texture addressers
# are and how many vector-memory instructions a product issues -- what bounds the pattern kernels (DESIGN 3.4b end, 8).
# Every counter set in a pass of its own.  usage: pmc_short_rows.sh [TAG]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05/pmc_short_rows}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=${PMC_SETS:-"GRBM_GUI_ACTIVE,TA_BUSY_avr SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_WAVES SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum FETCH_SIZE"}
one() {   # name, env ("A=1,B=2" or "-"), spec
  local name=$1 envs=$2 spec=$3
  echo "== $name: $spec ${envs}"
  for CS in $SETS; do
    local C=$(echo $CS | tr "," " ") T=$(echo $CS | tr ',' '_')
    local d=$OUT/$name/$T
    rm -rf "$d"; mkdir -p "$d"
    if [ "$envs" != "-" ]; then for e in ${envs//,/ }; do export $e; done; fi
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$d" -o p -- python3 "$R/bench.py" --workload $spec --roofline-only > "$d/out.txt" 2> "$d/err.txt"
    if [ "$envs" != "-" ]; then for e in ${envs//,/ }; do unset ${e%%=*}; done; fi
    local F=$(find "$d" -name "*counter_collection.csv" | head -1)
    [ -z "$F" ] && { echo "  $T: no counters"; tail -2 "$d/err.txt"; continue; }
    python3 - "$F" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_spmv_" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(by.items()):
    v = v[3:] if len(v) > 6 else v          # (the first launches warm up)
    print(f"  {k:60s} {c:32s} per launch {sum(v) / len(v):.6g}   [n={len(v)}]")
PY
  done
}
one wide_mesh - mesh2d:4000:4000:16:16
one one_byte_patterns - poisson2d:4000:4000
one packed_records LSQRHIP_PAT=0 poisson2d:4000:4000
one structure_patterns LSQRHIP_PAT=0,LSQRHIP_VAL8=0 poisson2d:4000:4000
