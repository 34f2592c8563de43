#!/bin/bash
# BASELINE.json configurations that fit ONE MI355X, through bench.py (one JSON line each).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo "== $1"; timeout 900 python bench.py --steps $2 --warmup $3 --cpu-iters 0 --no-scaling-ref --workload $1 2>/dev/null | tee -a gpurun_out/bench_configs.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; k=d['kernels']
print('  %.1f it/s  %.3f ms/it | spmv1 %.1f us %.0f GB/s (%.1f%%) | spmv2 %.1f us %.0f GB/s | update %.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['achieved'], 100*r['frac'], k['spmv_mode2']['avg_launch_us'], k['spmv_mode2']['gbps'], k['update_xw']['avg_launch_us']))"; }
rm -f gpurun_out/bench_configs.jsonl
run poisson2d:1000:1000 2000 200
run random:4000000:1000000:1000 20 2
run random:4000000:1000000:100 40 4
run random:10000000:10000000:100 40 4
run powerlaw:5000000:2000000:10000 100 10
