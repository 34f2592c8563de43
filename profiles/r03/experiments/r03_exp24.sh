#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['kernels']['update_xw']['avg_launch_us'],2))"; }
{
for r in 1 2; do
for cfg in "2 1024" "2 1536" "1 1536" "2 2048" "3 1024"; do
set -- $cfg
LSQRHIP_PAT_U=$1 LSQRHIP_SELL_GRID=$2 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=$1 grid=$2 poisson4000"
done
done
for cfg in "2 1024" "2 896" "2 1152" "3 1024" "3 768"; do
set -- $cfg
LSQRHIP_PAT_U=$1 LSQRHIP_SELL_GRID=$2 timeout 600 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=$1 grid=$2 config2"
done
} > gpurun_out/r03_exp24.txt 2>&1
