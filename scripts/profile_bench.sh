#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (separately, as the
# MI355X guide prescribes).  Usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 400 --warmup 40 --cpu-iters 0 --no-scaling-ref $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$R/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" $ARGS --no-roofline > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" $ARGS --no-roofline > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
find "$OUT" -name "*.csv" | head -20
python3 "$R/scripts/summarize_profile.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
