#!/bin/bash
# r06 evidence, ONE build, one gpurun call: the bench lines (default and the driver's K = 20 form), the N > 1 line shape at
# world 1, rocprofv3 summaries of every reported workload (scripts/profile_r06.sh), the RCCL-ranks-sharing-one-GPU lines of
# configs[3], PMC of configs[3] and [4], the rank-block series, the 18-problem suite on the device, the GPU suite
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
OUT=$R/gpurun_out/r06f
mkdir -p $OUT
( time python bench.py --detail $OUT/bench_detail.json ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json; echo
( time python bench.py --steps 20 --warmup 5 --detail $OUT/bench_detail_k20.json ) > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
LSQR_BENCH_FORCE_DIST=1 LSQR_BENCH_STRONG_REF=0 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --workload random:1250000:10000000:100 --steps 100 --warmup 10 --detail $OUT/engine_1rank_shard8_detail.json 2> $OUT/engine_1rank_shard8.err | grep '^{' | tail -1 > $OUT/engine_1rank_shard8.json
for ov in 0 1; do
  LSQR_BENCH_STRONG_REF=1 LSQRHIP_SHARD_OVERLAP=$ov LSQR_RANKS_SHARE_GPU=1 LSQR_DIST_PROBE_TIMEOUT=600 python bench.py --gpus 8 --steps 8 --warmup 2 --workload random:10000000:10000000:100 --traffic off --cpu-iters 0 --detail $OUT/rccl_shared_gpu_configs3_w8_overlap${ov}_detail.json 2> $OUT/rccl_shared_gpu_configs3_w8_overlap$ov.err | grep '^{' | tail -1 > $OUT/rccl_shared_gpu_configs3_w8_overlap$ov.json
done
LSQR_BENCH_STRONG_REF=0 LSQR_RANKS_SHARE_GPU=1 LSQR_DIST_PROBE_TIMEOUT=600 python bench.py --gpus 4 --steps 8 --warmup 2 --workload random:2000000:1000000:50 --traffic off --cpu-iters 0 --detail $OUT/rccl_shared_gpu_variants_w4_detail.json 2> $OUT/rccl_shared_gpu_variants_w4.err | grep '^{' | tail -1 > $OUT/rccl_shared_gpu_variants_w4.json
bash scripts/rank_block_times.sh $OUT/rank_block_times.txt > /dev/null 2>&1
bash scripts/profile_r06.sh config4 shard8 shard8_r1000 shard8_plan config3_100 config3_literal config5 config2 poisson4000_pat poisson4000_val8 mesh4000_wide > $OUT/profile_log.txt 2>&1
cp $R/gpurun_out/r06/*.txt $R/gpurun_out/r06/*.json $OUT/ 2>/dev/null
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE WRITE_SIZE"
bash scripts/pmc_csb.sh random:10000000:10000000:100 r06f/pmc_c4 > $OUT/pmc_config4.txt 2>&1
bash scripts/pmc_csb.sh random:1250000:10000000:100 r06f/pmc_s8 > $OUT/pmc_shard8.txt 2>&1
bash scripts/pmc_csb.sh powerlaw:5000000:2000000:10000 r06f/pmc_c5 > $OUT/pmc_config5.txt 2>&1
rm -rf $OUT/pmc_c4 $OUT/pmc_c5 $OUT/pmc_s8
for spec in random:1250000:10000000:100 powerlaw:5000000:2000000:10000 random:10000000:10000000:100; do python3 scripts/csb_probe.py $spec 2>&1 | grep -v amdgpu.ids; done > $OUT/csb_phase_clocks.txt
python -m lsqr_amd.operator > $OUT/LSQR_gpu_mi355x.LIS 2> $OUT/LSQR_gpu.err
python -m pytest tests -m gpu -q -rf --durations=15 2>&1 | tail -60 > $OUT/full_gpu.txt
tail -3 $OUT/full_gpu.txt
