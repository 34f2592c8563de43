#!/bin/bash
# y stores of the pattern kernel: plain, nontemporal, write-through (sc0 sc1)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2))"; }
{
for r in 1 2; do
for lib in liblsqrhip.so liblsqrhip_ys1.so liblsqrhip_ys2.so; do
LSQRHIP_LIB=$lib LSQRHIP_PAT_U=1 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib U=1 K=2000"
LSQRHIP_LIB=$lib LSQRHIP_PAT_U=2 LSQRHIP_SELL_GRID=1024 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib U=2 grid=1024 K=2000"
done
done
for lib in liblsqrhip.so liblsqrhip_ys1.so liblsqrhip_ys2.so; do
LSQRHIP_LIB=$lib LSQRHIP_PAT_U=1 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib U=1 poisson4000"
done
} > gpurun_out/r03_exp20.txt 2>&1
