#!/bin/bash
# r03 experiment 6: SELLP_EARLY=2, engine tests, the N>1 bench line at world = 1, sharded initialise time
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=$R/gpurun_out/r03_exp6.txt
mkdir -p gpurun_out
{
echo "### engine + fortran tests"
timeout 1800 python -m pytest tests/test_gpu_engine.py tests/test_fortran.py -q -x 2>&1 | tail -15
echo "### config 2: default vs early2"
for r in 1 2; do for lib in liblsqrhip.so liblsqrhip_early2.so liblsqrhip_head.so; do
  for K in 2000 20; do echo "$lib K=$K"; LSQRHIP_LIB=$lib timeout 300 python bench.py --steps $K --warmup $((K/10)) --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'], d['kernels']['spmv_mode2']['avg_launch_us'])"; done
done; done
echo "### N>1 line forced at world = 1 (one rank's block of config 4 at N = 8)"
LSQR_BENCH_FORCE_DIST=1 timeout 900 python bench.py --gpus 1 --steps 40 --warmup 4 --workload random:1250000:10000000:100 > gpurun_out/engine_1rank_shard8.json 2> gpurun_out/engine_1rank_shard8.err; tail -c 3000 gpurun_out/engine_1rank_shard8.json
echo "### sharded initialise, one pass (new) vs one pass per rank (head)"
timeout 900 python scripts/sharded_init_time.py random:2000000:1000000:50 1 8
LSQRHIP_LIB=liblsqrhip_head.so timeout 900 python scripts/sharded_init_time.py random:2000000:1000000:50 1 8
} > $O 2>&1
tail -5 $O
