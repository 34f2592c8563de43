#!/bin/bash
# what ONE rank of the strong-scaling series does per iteration, exchanges apart: the sharded engine at world = 1 on the
# row block a rank of an N-GPU run of configs[3] holds (N = 1, 2, 4, 8).  usage: rank_block_times.sh OUT
cd ${GRAFT_REPO_ROOT:-.}
OUT=$1
for rows in 10000000 5000000 2500000 1250000; do
  LSQR_BENCH_FORCE_DIST=1 LSQR_BENCH_STRONG_REF=0 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload random:$rows:10000000:100 --steps 60 --warmup 6 --traffic off --cpu-iters 0 --detail /tmp/rank_block_detail.json 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('rows per rank $rows: engine at world = 1 %.4f ms per iteration | mode 1 %.1f us (frac %.3f) mode 2 %.1f us (frac %.3f)  [fractions on SURVEY 8d bytes]' % (d['ms_per_step'], r['avg_launch_us'], r['frac'], r['avg_launch_us_mode2'], r['frac_mode2']))" | tee -a $OUT
done
