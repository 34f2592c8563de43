#!/bin/bash
# write-through stores for y only: plain build against the default one; pattern and packed layouts; grids
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['kernels']['update_xw']['avg_launch_us'],2))"; }
{
for r in 1 2; do
for lib in liblsqrhip_st0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib LSQRHIP_PAT_U=1 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib pat U=1 K=2000"
LSQRHIP_LIB=$lib LSQRHIP_PAT_U=2 LSQRHIP_SELL_GRID=1024 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib pat U=2 grid=1024 K=2000"
LSQRHIP_LIB=$lib LSQRHIP_PAT=0 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib sellp K=2000"
done
done
for g in 768 1280 1536; do
LSQRHIP_PAT_U=2 LSQRHIP_SELL_GRID=$g timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "default pat U=2 grid=$g K=2000"
done
for g in 1024 1280; do
LSQRHIP_PAT_U=1 LSQRHIP_SELL_GRID=$g timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "default pat U=1 grid=$g K=2000"
LSQRHIP_PAT=0 LSQRHIP_SELL_GRID=$g timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "default sellp grid=$g K=2000"
done
LSQRHIP_PAT_U=1 python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
LSQRHIP_PAT_U=2 LSQRHIP_SELL_GRID=1024 python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r03_exp22.txt 2>&1
