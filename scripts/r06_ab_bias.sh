#!/bin/bash
# r06: how much of an A/B of scripts/ab_env.py is the ORDER the variants were built in?  The same variant twice ("1" and "01":
# both read as 1), and the fused / combine pair in both orders.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
OUT=gpurun_out/r06/ab_harness_order_bias.txt
: > $OUT
for spec in random:1250000:10000000:100 random:10000000:10000000:100; do
  timeout 900 python3 scripts/ab_env.py $spec LSQRHIP_CSB_FUSE=1,01,001 10 5 2>&1 | tail -3 | tee -a $OUT
done
timeout 900 python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_FUSE=0,1 10 5 2>&1 | tail -2 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_FUSE=1,0 10 5 2>&1 | tail -2 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py powerlaw:5000000:2000000:10000 LSQRHIP_CSB_FUSE=0,1 10 5 2>&1 | tail -2 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py powerlaw:5000000:2000000:10000 LSQRHIP_CSB_FUSE=1,0 10 5 2>&1 | tail -2 | tee -a $OUT
