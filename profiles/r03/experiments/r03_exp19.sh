#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2))"; }
{
for u in 1 2 3 4; do
for g in 512 768 1024 1280 1536; do
LSQRHIP_PAT_U=$u LSQRHIP_SELL_GRID=$g timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=$u grid=$g K=2000"
done
done
for u in 2 3 4; do
for g in 768 1024; do
LSQRHIP_PAT_U=$u LSQRHIP_SELL_GRID=$g python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
LSQRHIP_PAT_U=$u LSQRHIP_SELL_GRID=$g timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=$u grid=$g poisson4000"
done
done
} > gpurun_out/r03_exp19.txt 2>&1
