#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06/csb_probe.txt
mkdir -p gpurun_out/r06
: > $OUT
python -m pytest tests/test_gpu_csb.py -x -q 2>&1 | tail -2 | tee -a $OUT
for spec in random:1250000:10000000:100 powerlaw:5000000:2000000:10000; do
  timeout 300 python3 scripts/csb_probe.py $spec 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
for spec in random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:2500000:10000000:100; do
  timeout 600 python3 scripts/ab_env.py $spec LSQRHIP_CSB_FUSE=1,0 10 5 2>&1 | tail -2 | tee -a $OUT
done
LSQRHIP_SHARD_OVERLAP=1 LSQRHIP_SHARD_WORLD=8 timeout 600 python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_FUSE=1,0 10 5 2>&1 | tail -2 | sed 's/^/overlap plan (P = 8, G = 2): /' | tee -a $OUT
