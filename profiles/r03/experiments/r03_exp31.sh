#!/bin/bash
# k_start: the zeros and u = b stored through to memory (default build) against plain stores (liblsqrhip_t0.so)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for r in 1 2 3; do
for lib in liblsqrhip_t0.so liblsqrhip.so; do
LSQRHIP_LIB=$lib python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
done
done
export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl -o t -- python3 scripts/short_solve_timeline.py 2>&1 | grep "wall us"
python3 scripts/short_solve_timeline.py --parse /tmp/tl 2>&1 | grep -v "k_spmv" | head -30
} > gpurun_out/r03_exp31.txt 2>&1
