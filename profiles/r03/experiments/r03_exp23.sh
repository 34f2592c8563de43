#!/bin/bash
# write-through y stores per launch kind: 1 = only the plain product (mode 2 in the loop), 2 = only the launch that carries the update
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['kernels']['update_xw']['avg_launch_us'],2))"; }
{
for r in 1 2; do
for lib in liblsqrhip.so liblsqrhip_y1.so liblsqrhip_y2.so; do
LSQRHIP_LIB=$lib LSQRHIP_PAT_U=2 LSQRHIP_SELL_GRID=1024 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib pat U=2 grid=1024 K=2000"
LSQRHIP_LIB=$lib LSQRHIP_PAT=0 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib sellp K=2000"
done
done
} > gpurun_out/r03_exp23.txt 2>&1
