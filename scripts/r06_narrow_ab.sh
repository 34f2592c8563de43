#!/bin/bash
# r06: the 11-byte index stream (u16 local row + u8 column delta, LSQRHIP_CSB_NARROW=1) under the lock-step sweep.  Round 4
# measured it equal to the 12-byte form under the free-running sweep; the lock step runs at the lines a CU keeps in flight,
# where 8 % fewer stream lines could show.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
OUT=gpurun_out/r06/narrow_ab.txt
: > $OUT
for spec in random:10000000:10000000:100 random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:4000000:1000000:100; do
  timeout 900 python3 scripts/ab_env.py $spec LSQRHIP_CSB_NARROW=0,1 10 5 2>&1 | tail -2 | tee -a $OUT
done
