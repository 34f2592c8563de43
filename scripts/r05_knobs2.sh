#!/bin/bash
# round 5, after the single-barrier lock step: rounds, narrow stream, K, splits once more; PMC of config 4 with and without splits
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/lockstep_knobs2.txt
: > $OUT
for spec in random:10000000:10000000:100 random:1250000:10000000:100 powerlaw:5000000:2000000:10000; do
  timeout 600 python3 scripts/ab_env.py $spec LSQRHIP_CSB_ROUNDS=1,0 5 3 2>&1 | tail -2 | tee -a $OUT
  timeout 600 python3 scripts/ab_env.py $spec LSQRHIP_CSB_NARROW=0,1 5 3 2>&1 | tail -2 | tee -a $OUT
done
timeout 900 python3 scripts/ab_env.py random:10000000:10000000:100 LSQRHIP_CSB_S=1,2,4 5 3 2>&1 | tail -3 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_S=-,2,4 5 3 2>&1 | tail -3 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:5000000:10000000:100 LSQRHIP_CSB_S=1,2 5 3 2>&1 | tail -2 | tee -a $OUT
timeout 900 python3 scripts/ab_env.py random:2500000:10000000:100 LSQRHIP_CSB_S=-,1,2 5 3 2>&1 | tail -3 | tee -a $OUT
PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE" bash scripts/pmc_csb.sh random:10000000:10000000:100 r05/pmc_c4_s1 2>&1 | tee gpurun_out/r05/pmc_config4_s1.txt
LSQRHIP_CSB_S=4 PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE" bash scripts/pmc_csb.sh random:10000000:10000000:100 r05/pmc_c4_s4 2>&1 | tee gpurun_out/r05/pmc_config4_s4.txt
rm -rf gpurun_out/r05/pmc_c4_s1 gpurun_out/r05/pmc_c4_s4
