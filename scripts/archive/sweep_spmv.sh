#!/bin/bash
# sweep SpMV tuning knobs over a few workloads (run on the GPU box)
cd "$(dirname "$0")/.."
for spec in poisson2d:1000:1000 random:1000000:1000000:20 random:400000:100000:100 powerlaw:500000:200000:10000; do
  for C in 512 1024 2048 4096; do
    for EVEN in 0 1; do
      LSQRHIP_SPMV_C=$C LSQRHIP_SPMV_EVEN=$EVEN timeout 120 python scripts/kernel_times.py $spec 200 2>/dev/null
    done
  done
  LSQRHIP_SPMV_C=1024 LSQRHIP_SPMV_EVEN=1 LSQRHIP_SPMV_WGS=4 timeout 120 python scripts/kernel_times.py $spec 200 2>/dev/null
  LSQRHIP_SPMV_C=2048 LSQRHIP_SPMV_EVEN=1 LSQRHIP_SPMV_WGS=4 timeout 120 python scripts/kernel_times.py $spec 200 2>/dev/null
done
