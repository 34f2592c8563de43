#!/bin/bash
# pattern kernel: first trip's requests ahead of the prologue (default build) against behind it (liblsqrhip_e0.so)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['kernels']['update_xw']['avg_launch_us'],2))"; }
{
echo "### tests"
timeout 900 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for r in 1 2 3; do
for lib in liblsqrhip_e0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib K=2000"
done
done
for lib in liblsqrhip_e0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
LSQRHIP_LIB=$lib timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib poisson4000"
LSQRHIP_LIB=$lib LSQRHIP_PAT_GRID=1536 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib grid=1536 K=2000"
LSQRHIP_LIB=$lib LSQRHIP_PAT_GRID=768 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib grid=768 K=2000"
done
} > gpurun_out/r03_exp26.txt 2>&1
