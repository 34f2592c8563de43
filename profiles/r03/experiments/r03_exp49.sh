#!/bin/bash
# column-swept row blocks on matrices that fit the Infinity Cache: plain loads of the stream (liblsqrhip_cp.so) against non-temporal
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for spec in random:1000000:500000:8 random:1000000:500000:16 random:2000000:1000000:16 random:4000000:1000000:32; do
for lib in liblsqrhip.so liblsqrhip_cp.so; do
LSQRHIP_LIB=$lib python3 scripts/kernel_times.py $spec 100 2>&1 | grep -v amdgpu.ids
LSQRHIP_LIB=$lib timeout 300 python bench.py --workload $spec --steps 200 --warmup 20 --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   solve it/s', round(d['value'],1), d['roofline']['kernel'])"
done
done
} > gpurun_out/r03_exp49.txt 2>&1
