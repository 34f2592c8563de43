#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
python3 scripts/k20_start_probe.py 2>&1 | grep -v amdgpu.ids
K=2000 python3 scripts/k20_start_probe.py 2>&1 | grep -v amdgpu.ids
echo "### parity subset with start_eager=1"
LSQRHIP_START_EAGER=1 LSQRHIP_LOOP_EVENTS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_range.py -q -x 2>&1 | tail -3
} > gpurun_out/r03_exp14.txt 2>&1
