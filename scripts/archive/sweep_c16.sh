#!/bin/bash
cd "$(dirname "$0")/.."
for C in 0 1 0 1; do LSQRHIP_COL16=$C timeout 120 python scripts/kernel_times.py poisson2d:1000:1000 400 2>/dev/null; done
