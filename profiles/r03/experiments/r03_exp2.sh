#!/bin/bash
# r03 experiment 2: pacing of the sweeps (LSQRHIP_CSB_PACE = steps a wave may lead the slowest workgroup of its XCD group)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_exp2.txt
mkdir -p gpurun_out
{
C4=random:10000000:10000000:100
for P in 0 1 2 3 4 6 8 16 64; do LSQRHIP_CSB_PACE=$P timeout 300 python scripts/kernel_times.py $C4 10; done
for s in random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:4000000:1000000:100; do
  for P in 0 2 4 8; do LSQRHIP_CSB_PACE=$P timeout 300 python scripts/kernel_times.py $s 10; done
done
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE"
for P in 2 4 8; do LSQRHIP_CSB_PACE=$P timeout 600 bash scripts/pmc_csb.sh $C4 pmc_c4_pace$P; done
} > $O 2>&1
tail -40 $O
