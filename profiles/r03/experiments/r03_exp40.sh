#!/bin/bash
# structure patterns (sell = 4): tests, fuzz, and a variable-coefficient stencil against sliced ELL (LSQRHIP_SPAT=0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['roofline']['frac'],3), d['roofline']['bytes_per_launch'])"; }
{
echo "### tests"
timeout 1500 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_formats.py tests/test_gpu_range.py tests/test_gpu_real32.py -q -x 2>&1 | tail -12
echo "### fuzz"
timeout 900 python scripts/fuzz_layouts.py 60 51 2>&1 | tail -5
echo "### 8-byte values at 1M and 16M rows: structure patterns against sliced ELL"
for spat in 0 1; do
LSQRHIP_PAT=0 LSQRHIP_VAL8=0 LSQRHIP_SPAT=$spat timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "SPAT=$spat 1M"
LSQRHIP_PAT=0 LSQRHIP_VAL8=0 LSQRHIP_SPAT=$spat timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "SPAT=$spat 16M"
done
} > gpurun_out/r03_exp40.txt 2>&1
