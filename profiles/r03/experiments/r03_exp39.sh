#!/bin/bash
# pattern kernel at 16M rows: y and the pattern numbers loaded non-temporally (liblsqrhip_nt.so) against plain loads
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2))"; }
{
for r in 1 2 3; do
for lib in liblsqrhip.so liblsqrhip_nt.so; do
LSQRHIP_LIB=$lib timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib poisson4000"
done
done
LSQRHIP_LIB=liblsqrhip_nt.so timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "nt config2"
} > gpurun_out/r03_exp39.txt 2>&1
