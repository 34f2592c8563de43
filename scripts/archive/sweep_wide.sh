#!/bin/bash
cd "$(dirname "$0")/.."
for spec in poisson2d:1000:1000 random:1000000:1000000:20 random:4000000:1000000:100 powerlaw:5000000:2000000:10000; do
  for W in 0 1; do
    LSQRHIP_SPMV_WIDE=$W timeout 300 python scripts/kernel_times.py $spec 200 2>/dev/null
  done
done
