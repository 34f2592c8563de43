#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_exp11.txt
{
C4=random:10000000:10000000:100
timeout 300 python scripts/kernel_times.py $C4 10
LSQRHIP_CSB_ROUNDS=0 timeout 300 python scripts/kernel_times.py $C4 10
timeout 300 python scripts/kernel_times.py $C4 10
LSQRHIP_CSB_ROUNDS=0 timeout 300 python scripts/kernel_times.py $C4 10
SH=random:1250000:10000000:100
for S in 0 2 8; do LSQRHIP_CSB_S=$S timeout 300 python scripts/kernel_times.py $SH 10; done
} > $O 2>&1
cat $O
