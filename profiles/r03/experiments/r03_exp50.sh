#!/bin/bash
# workgroups of the row-pattern kernel at 4M and 16M rows (the 1M-row choice is 1024)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['roofline']['frac'],3))"; }
{
for r in 1 2; do
for g in 1024 1536 2048; do
LSQRHIP_PAT_GRID=$g timeout 600 python bench.py --workload poisson2d:2000:2000 --steps 400 --warmup 40 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "4M grid=$g"
LSQRHIP_PAT_GRID=$g timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "16M grid=$g"
done
done
} > gpurun_out/r03_exp50.txt 2>&1
