#!/bin/bash
# paired rows: workgroups of the launch at config 2 (alternating processes)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
{
for i in 1 2; do for g in 1024 768 1536 2048; do
  for K in 2000 20; do
    v=$(LSQRHIP_PAT_GRID=$g python bench.py --steps $K --warmup 5 --extras off --traffic off --cpu-iters 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "LSQRHIP_PAT_GRID=$g  K = $K: $v it/s"
  done
done; done
} 2>&1 | tee gpurun_out/r05/pair_grid.txt
