#!/bin/bash
# round 5: the host's wait for a batch -- spin on the snapshot's seal (default) against hipEventSynchronize -- on the driver's
# K = 20 form of bench.py, alternating processes
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/spin_poll_ab.txt
: > $OUT
for i in 1 2 3 4; do
  for sp in 0 1; do
    v=$(LSQRHIP_SPIN_POLL=$sp python bench.py --steps 20 --warmup 5 --extras off --traffic off --cpu-iters 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "LSQRHIP_SPIN_POLL=$sp  K = 20: $v it/s" | tee -a $OUT
  done
done
for sp in 0 1; do
  v=$(LSQRHIP_SPIN_POLL=$sp python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
  echo "LSQRHIP_SPIN_POLL=$sp  K = 2000: $v it/s" | tee -a $OUT
done
