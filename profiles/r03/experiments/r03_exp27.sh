#!/bin/bash
# pattern kernel: the update between the requests and the sums of the first trip (default build) against ahead of the trips (liblsqrhip_s0.so)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['kernels']['update_xw']['avg_launch_us'],2))"; }
{
echo "### tests"
timeout 900 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for r in 1 2 3; do
for lib in liblsqrhip_s0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib K=2000"
done
done
for lib in liblsqrhip_s0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r03_exp27.txt 2>&1
