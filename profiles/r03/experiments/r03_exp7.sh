#!/bin/bash
# r03 experiment 7: the whole GPU suite
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_exp7.txt
mkdir -p gpurun_out
{
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -40
} > $O 2>&1
tail -5 $O
