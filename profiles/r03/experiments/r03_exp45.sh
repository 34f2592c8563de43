#!/bin/bash
# column splits: the splits' integer sums stored / loaded non-temporally (liblsqrhip_znt.so) against plain
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for r in 1 2; do
for lib in liblsqrhip.so liblsqrhip_znt.so; do
LSQRHIP_LIB=$lib python3 scripts/kernel_times.py random:10000000:10000000:100 20 2>&1 | grep -v amdgpu.ids
LSQRHIP_LIB=$lib python3 scripts/kernel_times.py random:1250000:10000000:100 40 2>&1 | grep -v amdgpu.ids
done
done
} > gpurun_out/r03_exp45.txt 2>&1
