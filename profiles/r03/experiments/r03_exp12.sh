#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for G in 1024 1280 1536 1792 2048; do
  for r in 1 2; do
  echo -n "SELL_GRID=$G K=2000: "; LSQRHIP_SELL_GRID=$G timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['avg_launch_us'],2), round(d['kernels']['spmv_mode2']['avg_launch_us'],2))"
  done
done
