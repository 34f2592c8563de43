#!/bin/bash
# r03 experiment 5: how the (value, index) stream is loaded / allocated vs L2 residency of x; engine tests
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=$R/gpurun_out/r03_exp5.txt
mkdir -p gpurun_out
{
echo "### engine tests"
timeout 1800 python -m pytest tests/test_gpu_engine.py -q -x 2>&1 | tail -30
echo "### stream policies on config 4"
C4=random:10000000:10000000:100
for lib in liblsqrhip.so liblsqrhip_plain.so liblsqrhip_sys.so liblsqrhip_agent.so; do
  echo "== $lib"; LSQRHIP_LIB=$lib timeout 300 python scripts/kernel_times.py $C4 10
done
echo "== nt + uncached allocation"; LSQRHIP_CSB_UC=1 timeout 300 python scripts/kernel_times.py $C4 10
echo "== plain + uncached allocation"; LSQRHIP_LIB=liblsqrhip_plain.so LSQRHIP_CSB_UC=1 timeout 300 python scripts/kernel_times.py $C4 10
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE"
echo "### PMC: sys"; LSQRHIP_LIB=liblsqrhip_sys.so timeout 600 bash scripts/pmc_csb.sh $C4 pmc_c4_sys
echo "### PMC: nt + UC"; LSQRHIP_CSB_UC=1 timeout 600 bash scripts/pmc_csb.sh $C4 pmc_c4_uc
} > $O 2>&1
tail -5 $O
