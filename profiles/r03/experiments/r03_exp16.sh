#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
echo "### pattern tests"
timeout 900 python -m pytest tests/test_gpu_patterns.py -q -x 2>&1 | tail -15
echo "### config 2"
for pat in 0 1; do
LSQRHIP_PAT=$pat timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PAT=$pat', round(d['value']), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2), round(d['roofline']['frac'],3), d['roofline']['bytes_per_launch'])"
LSQRHIP_PAT=$pat python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
done
echo "### fuzz"
timeout 900 python scripts/fuzz_layouts.py 60 21 2>&1 | tail -6
} > gpurun_out/r03_exp16.txt 2>&1
