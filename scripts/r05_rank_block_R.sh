#!/bin/bash
# one rank's block of configs[3] at N = 8: column splits (the build's choice) against smaller row blocks without splits
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
{
python scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_S=-,1,2,4 10 5
LSQRHIP_CSB_S=1 python scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_R=20352,10176,5088,4096 10 5
LSQRHIP_CSB_S=2 python scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_R=20352,10176 10 5
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/rank_block_R_vs_splits.txt
