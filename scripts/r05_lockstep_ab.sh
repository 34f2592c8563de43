#!/bin/bash
# round 5: the lock-step sweep (csb.h) against the free-running one, library kernels, one process per shape
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/lockstep_ab.txt
: > $OUT
for spec in random:10000000:10000000:100 random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:4000000:1000000:100; do
  timeout 600 python3 scripts/ab_env.py $spec LSQRHIP_CSB_LOCKSTEP=0,1,2 5 4 2>&1 | tail -3 | tee -a $OUT
done
