#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_exp10.txt
{
timeout 900 python -m pytest tests/test_gpu_csb.py tests/test_gpu_fullsize_parity.py tests/test_gpu_devgen.py -x -q 2>&1 | tail -4
for s in random:10000000:10000000:100 random:1250000:10000000:100 random:4000000:1000000:100 powerlaw:5000000:2000000:10000 random:4000000:1000000:1000 random:20000000:10000000:30; do timeout 600 python scripts/kernel_times.py $s 10; done
LSQRHIP_CSB_S=2 timeout 300 python scripts/kernel_times.py random:10000000:10000000:100 10
LSQRHIP_CSB_S=1 timeout 300 python scripts/kernel_times.py random:10000000:10000000:100 10
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE"
LSQRHIP_CSB_S=2 timeout 600 bash scripts/pmc_csb.sh random:10000000:10000000:100 pmc_c4_S2
} > $O 2>&1
tail -30 $O
