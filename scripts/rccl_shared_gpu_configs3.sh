#!/bin/bash
# BASELINE configs[3] as stated -- 10M x 10M, 100 per row, EIGHT ranks over RCCL -- on a ONE-GPU box: the ranks share
# device 0 (LSQR_RANKS_SHARE_GPU=1: one NCCL_HOSTID per rank, socket transport on lo).  Not a measurement of anything
# but correctness: the C++ engine's RCCL branch at world = 8 on the full-size problem, checked by dist_bench against the
# Python stage driver, and its result against one handle holding the whole matrix (the line's strong-scaling reference).
# usage: rccl_shared_gpu_configs3.sh OUTDIR [overlap=0|1] [world=8] [steps=20]
cd ${GRAFT_REPO_ROOT:-.}
OUT=$1; OV=${2:-0}; W=${3:-8}; K=${4:-20}
mkdir -p $OUT
export LSQR_RANKS_SHARE_GPU=1 LSQRHIP_SHARD_OVERLAP=$OV LSQR_BENCH_STRONG_REF=1 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT
timeout 1500 python bench.py --gpus $W --steps $K --warmup 2 --workload random:10000000:10000000:100 --traffic off --cpu-iters 0 \
  > $OUT/configs3_w${W}_ov${OV}.json 2> $OUT/configs3_w${W}_ov${OV}.err
echo "rc=$?"
grep '^{' $OUT/configs3_w${W}_ov${OV}.json | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','n_gpus','steps','ms_per_step','overlap','result')})
print(d['config']['engine'], d['config']['engine_note'], d['config']['backend'])
print('one handle, same workload, same iterations:', d['strong_scaling_ref'].get('result'), d['strong_scaling_ref'].get('sharded_vs_1gpu'))
"
grep -h "NCCL INFO" $OUT/configs3_w${W}_ov${OV}.json $OUT/configs3_w${W}_ov${OV}.err | grep -i "Init COMPLETE\|via NET\|Using network\|NCCL_HOSTID\|nranks\|NET/Socket" | cut -c1-220 | sort | uniq -c | sort -rn | head -40 > $OUT/configs3_w${W}_ov${OV}_nccl_info.txt
wc -l $OUT/configs3_w${W}_ov${OV}_nccl_info.txt
