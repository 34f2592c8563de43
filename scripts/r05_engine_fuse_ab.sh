#!/bin/bash
# round 5: the engine at world 1 on one rank's block, step 2 inside the update's launch (default) or in front of it
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/engine_fuse_s2_ab.txt
: > $OUT
for i in 1 2 3; do
 for f in 0 1; do
  v=$(LSQRHIP_SHARD_FUSE_S2=$f LSQR_BENCH_FORCE_DIST=1 LSQR_BENCH_STRONG_REF=0 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2957$f bench.py --gpus 1 --workload random:1250000:10000000:100 --steps 100 --warmup 10 --traffic off --cpu-iters 0 --no-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per iteration' % d['ms_per_step'])")
  echo "LSQRHIP_SHARD_FUSE_S2=$f engine at world 1: $v" | tee -a $OUT
 done
done
v=$(python bench.py --workload random:1250000:10000000:100 --extras off --steps 100 --warmup 10 --traffic off --cpu-iters 0 --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per iteration' % d['ms_per_step'])")
echo "one handle on the same block: $v" | tee -a $OUT
