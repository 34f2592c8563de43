#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for r in 1 2; do
for g in 0 128 256 512 1024; do
LSQRHIP_START_GRID=$g python3 scripts/k20_wall.py 2>&1 | grep -v amdgpu.ids
done
done
} > gpurun_out/r03_exp15.txt 2>&1
