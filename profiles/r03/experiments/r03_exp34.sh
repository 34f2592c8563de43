#!/bin/bash
# kernel breakdown of one rank's iteration in the sharded engine (world = 1, the N = 8 block of config 4)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export LSQR_BENCH_FORCE_DIST=1 LSQR_BENCH_STRONG_REF=0 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571
rm -rf /tmp/eng && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eng -o t -- python3 $R/bench.py --gpus 1 --workload random:1250000:10000000:100 --steps 100 --warmup 10 --traffic off --cpu-iters 0 > $R/gpurun_out/r03_exp34.json 2> $R/gpurun_out/r03_exp34.err
f=$(find /tmp/eng -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $R/gpurun_out/r03_exp34.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print("%-100s calls=%7s total_ns=%13s avg_ns=%11s pct=%6s" % (r.get("Name", "")[:100], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
PY
tail -c 600 $R/gpurun_out/r03_exp34.json >> $R/gpurun_out/r03_exp34.txt
