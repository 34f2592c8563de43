#!/bin/bash
# end of a batch: k_s3_snap and the copy of x as one launch (default build) against two (liblsqrhip_o0.so)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1))"; }
{
echo "### tests"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_patterns.py tests/test_gpu_range.py tests/test_gpu_real32.py -q -x 2>&1 | tail -3
for r in 1 2 3; do
for lib in liblsqrhip_o0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
done
done
for lib in liblsqrhip_o0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "$lib K=2000"
done
} > gpurun_out/r03_exp36.txt 2>&1
