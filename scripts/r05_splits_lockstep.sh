#!/bin/bash
# round 5: column splits under the lock-step sweep, the shapes the strong-scaling series holds
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/lockstep_splits_by_shape.txt
: > $OUT
timeout 900 python3 scripts/ab_env.py random:5000000:10000000:100 LSQRHIP_CSB_S=-,1,2,4 5 3 2>&1 | tail -4 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:2500000:10000000:100 LSQRHIP_CSB_S=-,1,2,4 5 3 2>&1 | tail -4 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_S=-,2,4,8 5 3 2>&1 | tail -4 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:10000000:10000000:100 LSQRHIP_CSB_S=-,1 5 3 2>&1 | tail -2 | tee -a $OUT
