#!/bin/bash
# A/B of the cross-block prefetch (spmv.h PF) x value dictionary (V8), interleaved on one box.
cd "$(dirname "$0")/.."
for s in poisson2d:1000:1000 poisson2d:2000:2000; do
  for V in 0 1; do for P in 0 1 0 1; do LSQRHIP_VAL8=$V LSQRHIP_PREFETCH=$P timeout 200 python scripts/kernel_times.py $s 300 2>/dev/null; done; done
done
for s in random:1000000:1000000:20 random:4000000:1000000:100 powerlaw:5000000:2000000:10000; do
  for P in 0 1 0 1; do LSQRHIP_PREFETCH=$P timeout 200 python scripts/kernel_times.py $s 100 2>/dev/null; done
done
