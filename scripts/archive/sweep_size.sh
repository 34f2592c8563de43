#!/bin/bash
# SpMV time vs problem size (fixed overhead vs streaming rate), value dictionary on/off.
cd "$(dirname "$0")/.."
for V in 1 0; do for n in 250 500 700 1000 1400 2000 2800 4000; do
  LSQRHIP_VAL8=$V timeout 200 python scripts/kernel_times.py poisson2d:$n:$n 300 2>/dev/null
done; done
