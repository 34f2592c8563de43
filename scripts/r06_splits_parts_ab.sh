#!/bin/bash
# r06, review items 2(c) and 4 with the fused splits in place: a rank of eight's block with 2 / 4 column splits in BOTH modes
# (mode 2: 10M rows -- the splits' partial sums are 160-320 MB on top of 1.7 GB); the overlap plan with 2 and 4 parts
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
OUT=gpurun_out/r06/splits_and_parts_ab.txt
: > $OUT
timeout 900 python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_S=-,2,4 10 5 2>&1 | tail -3 | tee -a $OUT
LSQRHIP_SHARD_OVERLAP=1 LSQRHIP_SHARD_WORLD=8 timeout 900 python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_SHARD_PARTS=2,4 10 5 2>&1 | tail -2 | sed 's/^/overlap plan, P = 8: /' | tee -a $OUT
