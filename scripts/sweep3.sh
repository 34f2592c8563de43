#!/bin/bash
cd "$(dirname "$0")/.."
for spec in poisson2d:1000:1000 random:400000:100000:100 powerlaw:500000:200000:10000; do
  for C in 512 1024 2048; do
    for MG in 0 8192 65536; do
      LSQRHIP_SPMV_C=$C LSQRHIP_SPMV_EVEN=0 LSQRHIP_SPMV_MAXGRID=$MG timeout 120 python scripts/kernel_times.py $spec 200 2>/dev/null
    done
  done
done
