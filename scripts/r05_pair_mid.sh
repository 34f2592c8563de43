#!/bin/bash
# paired rows: the middle entry of a (-1, 0, +1) run formed in the lane instead of gathered
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
{
python scripts/ab_env.py poisson2d:4000:4000 LSQRHIP_PAT_PAIR_MID=0,1 20 5
python scripts/ab_env.py poisson2d:1000:1000 LSQRHIP_PAT_PAIR_MID=0,1 200 5
for i in 1 2 3; do for mid in 0 1; do
  for K in 2000 20; do
    v=$(LSQRHIP_PAT_PAIR_MID=$mid python bench.py --steps $K --warmup 5 --extras off --traffic off --cpu-iters 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "LSQRHIP_PAT_PAIR_MID=$mid  K = $K: $v it/s"
  done
done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/pair_mid_ab.txt
