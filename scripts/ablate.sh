#!/bin/bash
cd "$(dirname "$0")/.."
for AB in 0 1 2 3 4 5 8 9 10 11 13; do
  LSQRHIP_ABLATE=$AB timeout 120 python scripts/kernel_times.py poisson2d:1000:1000 300 2>/dev/null
done
