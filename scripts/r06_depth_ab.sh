#!/bin/bash
# r06: the sweep's stream two steps ahead (LSQRHIP_CSB_LOCKSTEP=3) against one step ahead with 1 / 2 chunks per step
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06/depth_ab.txt
mkdir -p gpurun_out/r06
: > $OUT
python -m pytest tests/test_gpu_csb.py -x -q 2>&1 | tail -2 | tee -a $OUT
LSQRHIP_CSB_LOCKSTEP=3 python -m pytest tests/test_gpu_csb.py -x -q 2>&1 | tail -2 | tee -a $OUT
for spec in random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:10000000:10000000:100 random:4000000:1000000:100 random:1250000:10000000:1000; do
  timeout 900 python3 scripts/ab_env.py $spec LSQRHIP_CSB_LOCKSTEP=1,2,3 5 5 2>&1 | tail -3 | tee -a $OUT
done
