#!/bin/bash
# structure patterns: y stored through (liblsqrhip_yt.so) against plain stores; workgroups
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export LSQRHIP_PAT=0 LSQRHIP_VAL8=0
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2))"; }
{
for r in 1 2; do
for lib in liblsqrhip.so liblsqrhip_yt.so; do
LSQRHIP_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib 1M"
done
done
for g in 1024 1280 2048; do
LSQRHIP_SELL_GRID=$g timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "grid=$g 1M"
done
for lib in liblsqrhip.so liblsqrhip_yt.so; do
LSQRHIP_LIB=$lib timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib 16M"
done
} > gpurun_out/r03_exp41.txt 2>&1
