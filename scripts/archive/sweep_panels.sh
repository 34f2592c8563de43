#!/bin/bash
cd "$(dirname "$0")/.."
for spec in random:1000000:1000000:20 random:2000000:2000000:50 random:4000000:1000000:100 powerlaw:5000000:2000000:10000 random:10000000:10000000:100; do
  LSQRHIP_PANELS=0 timeout 300 python scripts/kernel_times.py $spec 30 2>/dev/null
  for KB in 1024 2560 3584; do
    LSQRHIP_PANELS=1 LSQRHIP_PANEL_KB=$KB timeout 300 python scripts/kernel_times.py $spec 30 2>/dev/null
  done
done
