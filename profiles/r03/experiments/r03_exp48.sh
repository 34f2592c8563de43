#!/bin/bash
# where does the stream policy flip?  2M, 4M and 8M rows with plain (LSQRHIP_STREAM_NT=0) and non-temporal (=1) streams
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['roofline']['frac'],3))"; }
{
for spec in poisson2d:1414:1414 poisson2d:2000:2000 poisson2d:2828:2828; do
for nt in 0 1; do
export LSQRHIP_STREAM_NT=$nt
timeout 600 python bench.py --workload $spec --steps 400 --warmup 40 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$spec NT=$nt pat"
LSQRHIP_PAT=0 timeout 600 python bench.py --workload $spec --steps 400 --warmup 40 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$spec NT=$nt packed"
LSQRHIP_PAT=0 LSQRHIP_VAL8=0 timeout 600 python bench.py --workload $spec --steps 400 --warmup 40 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$spec NT=$nt spat"
done
done
} > gpurun_out/r03_exp48.txt 2>&1
