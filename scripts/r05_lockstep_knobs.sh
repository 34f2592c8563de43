#!/bin/bash
# round 5: knobs of the lock-step sweep on the library kernels, one process per shape and knob
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/lockstep_knobs.txt
: > $OUT
for spec in random:10000000:10000000:100 random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:4000000:1000000:100; do
  timeout 600 python3 scripts/ab_env.py $spec LSQRHIP_CSB_STAGGER=0,2,4,6 5 4 2>&1 | tail -4 | tee -a $OUT
done
timeout 900 python3 scripts/ab_env.py random:10000000:10000000:100 LSQRHIP_CSB_S=1,2,4,8 5 3 2>&1 | tail -4 | tee -a $OUT
timeout 600 python3 scripts/ab_env.py powerlaw:5000000:2000000:10000 LSQRHIP_CSB_S=1,2 5 3 2>&1 | tail -2 | tee -a $OUT
timeout 600 python3 scripts/ab_env.py random:4000000:1000000:100 LSQRHIP_CSB_S=1,2 5 3 2>&1 | tail -2 | tee -a $OUT
