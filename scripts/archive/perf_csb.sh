#!/bin/bash
# column-swept row blocks (csb.h) against the L2 column panels on the scattered BASELINE shapes
cd ${GRAFT_REPO_ROOT:-.}
for spec in "$@"; do
  python scripts/kernel_times.py $spec 10
  LSQRHIP_CSB=0 python scripts/kernel_times.py $spec 10
done
