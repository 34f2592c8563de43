#!/bin/bash
# round 5: the second barrier of the lock step (in front of the gathers) by shape, library kernels, one process per shape
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/lockstep_barrier_a.txt
: > $OUT
for spec in random:4000000:1000000:100 random:1250000:10000000:1000 random:4000000:1000000:1000 random:10000000:10000000:100 random:1250000:10000000:100 powerlaw:5000000:2000000:10000; do
  timeout 900 python3 scripts/ab_env.py $spec LSQRHIP_CSB_BARRIER_A=0,1 4 3 2>&1 | tail -2 | tee -a $OUT
done
