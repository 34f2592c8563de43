#!/bin/bash
# r05 evidence: bench lines (default, the driver's K = 20 form), engine lines, rocprofv3 summaries (scripts/profile_r05.sh),
# the RCCL-ranks-sharing-one-GPU lines of configs[3], PMC of configs 4 and 5, the 18-problem suite on the device, GPU suite
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
OUT=$R/gpurun_out/r05
mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json
python bench.py --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
LSQR_BENCH_FORCE_DIST=1 LSQR_BENCH_STRONG_REF=0 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --workload random:1250000:10000000:100 --steps 100 --warmup 10 2> $OUT/engine_1rank_shard8.err | grep '^{' | tail -1 > $OUT/engine_1rank_shard8.json
LSQR_DIST_ENGINE=python LSQR_BENCH_FORCE_DIST=1 LSQR_BENCH_STRONG_REF=0 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 bench.py --gpus 1 --workload random:1250000:10000000:100 --steps 100 --warmup 10 --traffic off --cpu-iters 0 2> $OUT/engine_py_1rank_shard8.err | grep '^{' | tail -1 > $OUT/engine_py_1rank_shard8.json
python bench.py --workload random:1250000:10000000:100 --extras off --steps 100 --warmup 10 --traffic off --cpu-iters 0 > $OUT/one_handle_shard8.json 2> $OUT/one_handle_shard8.err
for ov in 0 1; do
  LSQR_BENCH_STRONG_REF=1 LSQRHIP_SHARD_OVERLAP=$ov LSQR_RANKS_SHARE_GPU=1 LSQR_DIST_PROBE_TIMEOUT=600 python bench.py --gpus 8 --steps 20 --warmup 2 --workload random:10000000:10000000:100 --traffic off --cpu-iters 0 2> $OUT/rccl_shared_gpu_configs3_w8_overlap$ov.err | grep '^{' | tail -1 > $OUT/rccl_shared_gpu_configs3_w8_overlap$ov.json
done
LSQR_BENCH_STRONG_REF=0 LSQR_RANKS_SHARE_GPU=1 LSQR_DIST_PROBE_TIMEOUT=600 python bench.py --gpus 4 --steps 20 --warmup 2 --workload random:2000000:1000000:50 --traffic off --cpu-iters 0 2> $OUT/rccl_shared_gpu_variants_w4.err | grep '^{' | tail -1 > $OUT/rccl_shared_gpu_variants_w4.json
bash scripts/profile_r05.sh config2 config2_packed poisson4000_pat poisson4000_dict poisson4000_val8 poisson4000_spat config4 shard8 shard8_r1000 shard8_plan config3_100 config3_literal config5 mesh4000_wide > $OUT/profile_log.txt 2>&1
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum TCP_TCC_READ_REQ_sum,TCP_TOTAL_CACHE_ACCESSES_sum FETCH_SIZE WRITE_SIZE"
bash scripts/pmc_csb.sh random:10000000:10000000:100 r05/pmc_c4 > $OUT/pmc_config4.txt 2>&1
bash scripts/pmc_csb.sh powerlaw:5000000:2000000:10000 r05/pmc_c5 > $OUT/pmc_config5.txt 2>&1
rm -rf $OUT/pmc_c4 $OUT/pmc_c5
python -m lsqr_amd.operator > $OUT/LSQR_gpu_mi355x.LIS 2> $OUT/LSQR_gpu.err
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/full_gpu.txt
tail -3 $OUT/full_gpu.txt
