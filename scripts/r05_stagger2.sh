#!/bin/bash
# round 5: stagger once more, on the final (single-barrier) lock step
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/lockstep_stagger_final.txt
: > $OUT
for spec in random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:10000000:10000000:100; do
  timeout 600 python3 scripts/ab_env.py $spec LSQRHIP_CSB_STAGGER=0,1,2,4 5 4 2>&1 | tail -4 | tee -a $OUT
done
