#!/bin/bash
# workgroups of the sell.h / structure-pattern kernels at 1M rows after the early share (LSQRHIP_SELL_GRID)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2))"; }
{
for g in 1024 1280 1536 1792; do
LSQRHIP_SELL_GRID=$g LSQRHIP_PAT=0 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "packed grid=$g"
LSQRHIP_SELL_GRID=$g LSQRHIP_PAT=0 LSQRHIP_VAL8=0 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "spat grid=$g"
LSQRHIP_SELL_GRID=$g LSQRHIP_PAT=0 LSQRHIP_VAL8=0 LSQRHIP_SPAT=0 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "sell8 grid=$g"
done
} > gpurun_out/r03_exp51.txt 2>&1
