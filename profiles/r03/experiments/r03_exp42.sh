#!/bin/bash
# 8-byte values loaded non-temporally (liblsqrhip_nt.so) in k_spmv_spat and k_spmv_sell, 16M and 1M rows
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export LSQRHIP_PAT=0 LSQRHIP_VAL8=0
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2))"; }
{
for r in 1 2 3; do
for lib in liblsqrhip.so liblsqrhip_nt.so; do
LSQRHIP_LIB=$lib timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib spat 16M"
done
done
for lib in liblsqrhip.so liblsqrhip_nt.so; do
LSQRHIP_SPAT=0 LSQRHIP_LIB=$lib timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib sell 16M"
LSQRHIP_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib spat 1M"
done
} > gpurun_out/r03_exp42.txt 2>&1
