#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['roofline']['frac'],3))"; }
{
echo "### pattern tests"
timeout 900 python -m pytest tests/test_gpu_patterns.py -q -x 2>&1 | tail -3
echo "### config 2"
for g in 1536 2048 1024; do
LSQRHIP_SELL_GRID=$g timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "grid=$g K=2000"
LSQRHIP_SELL_GRID=$g python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
done
echo "### 16M rows"
timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "poisson4000"
LSQRHIP_PAT=0 timeout 600 python bench.py --workload poisson2d:4000:4000 --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "poisson4000 PAT=0"
echo "### build time"
python3 scripts/build_time.py poisson2d:1000:1000 poisson2d:4000:4000 random:4000000:1000000:100 2>&1 | tail -5; LSQRHIP_PAT=0 python3 scripts/build_time.py poisson2d:1000:1000 poisson2d:4000:4000 2>&1 | tail -3
} > gpurun_out/r03_exp18.txt 2>&1
