#!/bin/bash
# r03 experiment 1: int64 accumulators (R = 20352) against r02's build on one box.
cd ${GRAFT_REPO_ROOT:-.}
L=lsqr_amd/lib
O=gpurun_out/r03_exp1.txt
mkdir -p gpurun_out
{
echo "### tests"
timeout 900 python -m pytest tests/test_gpu_csb.py -x -q 2>&1 | tail -5
cp $L/liblsqrhip.so /tmp/new.so
SPECS="random:10000000:10000000:100 random:1250000:10000000:100 random:4000000:1000000:100 powerlaw:5000000:2000000:10000"
echo "### new"
for s in $SPECS; do timeout 300 python scripts/kernel_times.py $s 10; done
echo "### head (r02)"
cp $L/liblsqrhip_head.so $L/liblsqrhip.so
for s in $SPECS; do timeout 300 python scripts/kernel_times.py $s 10; done
cp /tmp/new.so $L/liblsqrhip.so
echo "### new, column splits on config 4"
for S in 2 4 8; do LSQRHIP_CSB_S=$S timeout 300 python scripts/kernel_times.py random:10000000:10000000:100 10; done
echo "### new, one launch (no rounds)"
LSQRHIP_CSB_ROUNDS=0 timeout 300 python scripts/kernel_times.py random:10000000:10000000:100 10
echo "### new again"
for s in $SPECS; do timeout 300 python scripts/kernel_times.py $s 10; done
echo "### PMC new config 4"
timeout 900 bash scripts/pmc_csb.sh random:10000000:10000000:100 pmc_c4
echo "### PMC new config 4, 8 splits"
LSQRHIP_CSB_S=8 timeout 900 bash scripts/pmc_csb.sh random:10000000:10000000:100 pmc_c4_s8
} > $O 2>&1
tail -60 $O
