#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06/csb_probe.txt
mkdir -p gpurun_out/r06
: > $OUT
for spec in random:1250000:10000000:100 powerlaw:5000000:2000000:10000; do
  timeout 300 python3 scripts/csb_probe.py $spec 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  timeout 300 python3 scripts/csb_probe.py $spec LSQRHIP_CSB_FUSE=0 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
timeout 300 python3 scripts/csb_probe.py random:10000000:10000000:100 2>&1 | grep -v amdgpu.ids | tee -a $OUT
