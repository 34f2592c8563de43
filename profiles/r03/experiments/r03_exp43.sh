#!/bin/bash
# LSQRHIP_STREAM_NT = 0 / 1 (matrix stream loaded plainly / non-temporally) per layout, 16M and 1M rows
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['roofline']['frac'],3))"; }
{
echo "### tests (auto)"
timeout 1500 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_formats.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
LSQRHIP_STREAM_NT=1 timeout 1500 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_formats.py tests/test_gpu_real32.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for nt in 0 1; do
export LSQRHIP_STREAM_NT=$nt
for r in 1 2; do
timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "NT=$nt pat 16M"
LSQRHIP_PAT=0 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "NT=$nt packed 16M"
LSQRHIP_PAT=0 LSQRHIP_VAL8=0 LSQRHIP_SPAT=0 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "NT=$nt sell8 16M"
LSQRHIP_PAT=0 LSQRHIP_VAL8=0 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "NT=$nt spat 16M"
done
timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "NT=$nt pat 1M"
LSQRHIP_PAT=0 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "NT=$nt packed 1M"
done
unset LSQRHIP_STREAM_NT
timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "auto pat 1M"
LSQRHIP_PAT=0 LSQRHIP_VAL8=0 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "auto spat 16M"
} > gpurun_out/r03_exp43.txt 2>&1
