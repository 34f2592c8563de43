#!/bin/bash
# r03 experiment 9: column splits on the whole config 4 (each XCD then sweeps 1/S of x per launch)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_exp9.txt
{
C4=random:10000000:10000000:100
for S in 0 2 4 8; do LSQRHIP_CSB_S=$S timeout 300 python scripts/kernel_times.py $C4 10; done
C5=powerlaw:5000000:2000000:10000
for S in 0 2 4; do LSQRHIP_CSB_S=$S timeout 300 python scripts/kernel_times.py $C5 10; done
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE"
for S in 4 8; do LSQRHIP_CSB_S=$S timeout 600 bash scripts/pmc_csb.sh $C4 pmc_c4_S$S; done
} > $O 2>&1
tail -30 $O
