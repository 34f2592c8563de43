#!/bin/bash
# round 5: chunks per lock-step step (K = 1 / 2) on the dense-row shapes
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/lockstep_k_dense.txt
: > $OUT
timeout 900 python3 scripts/ab_env.py random:1250000:10000000:1000 LSQRHIP_CSB_LOCKSTEP=0,1,2 3 3 2>&1 | tail -3 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:4000000:1000000:1000 LSQRHIP_CSB_LOCKSTEP=1,2 3 3 2>&1 | tail -2 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:400000:100000:100 LSQRHIP_CSB_LOCKSTEP=0,1,2 10 3 2>&1 | tail -3 | tee -a $OUT
