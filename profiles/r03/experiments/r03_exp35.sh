#!/bin/bash
# sell.h kernels: this thread's share of the partial sums requested at the top (default build) against inside the prologue (liblsqrhip_h0.so)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export LSQRHIP_PAT=0
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['kernels']['update_xw']['avg_launch_us'],2))"; }
{
echo "### tests"
timeout 900 python -m pytest tests/test_gpu_formats.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for r in 1 2 3; do
for lib in liblsqrhip_h0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib packed K=2000"
LSQRHIP_VAL8=0 LSQRHIP_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib val8 K=2000"
done
done
} > gpurun_out/r03_exp35.txt 2>&1
