#!/bin/bash
# kernel stats of the sharded C++ engine at world = 1 on one rank's block of config 4 (what one rank of the N = 8 run does
# per iteration, minus the exchanges).  usage: prof_engine.sh [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-engine_w1}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LSQR_BENCH_FORCE_DIST=1 LSQR_BENCH_STRONG_REF=0 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$R/bench.py" --gpus 1 --workload ${SPEC:-random:1250000:10000000:100} --steps 100 --warmup 10 --traffic off --cpu-iters 0 --no-roofline > "$OUT/bench.json" 2> "$OUT/err.txt"
F=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
tail -1 "$OUT/bench.json" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('# value %.1f it/s ms_per_step %.4f engine %s' % (d['value'], d['ms_per_step'], d['config']['engine']))"
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-90s calls=%7s total_ns=%13s avg_ns=%11s pct=%6s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
PY
