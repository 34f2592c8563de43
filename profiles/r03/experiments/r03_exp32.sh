#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for v in 0 1; do
echo "## LSQRHIP_BSLOT_DEV=$v"
LSQRHIP_BSLOT_DEV=$v python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/tl; export LSQRHIP_BSLOT_DEV=$v; rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl -o t -- python3 scripts/short_solve_timeline.py 2>&1 | grep "wall us"
python3 scripts/short_solve_timeline.py --parse /tmp/tl 2>&1 | grep -v "k_spmv" | head -8
done
} > gpurun_out/r03_exp32.txt 2>&1
