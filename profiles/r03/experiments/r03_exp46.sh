#!/bin/bash
# y of the products stored non-temporally at HBM-resident sizes (liblsqrhip_ynt.so) against the default build
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['roofline']['frac'],3))"; }
{
for r in 1 2; do
for lib in liblsqrhip.so liblsqrhip_ynt.so; do
export LSQRHIP_LIB=$lib
timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib pat 16M"
LSQRHIP_PAT=0 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib packed 16M"
LSQRHIP_PAT=0 LSQRHIP_VAL8=0 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib spat 16M"
done
done
} > gpurun_out/r03_exp46.txt 2>&1
