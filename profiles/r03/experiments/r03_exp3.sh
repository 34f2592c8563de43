#!/bin/bash
# r03 experiment 3: pacing (second form), early loads of the packed-SELL kernel (config 2), new tests
cd ${GRAFT_REPO_ROOT:-.}
L=lsqr_amd/lib
O=gpurun_out/r03_exp3.txt
mkdir -p gpurun_out
{
echo "### new tests"
timeout 1200 python -m pytest tests/test_gpu_csb.py tests/test_gpu_formats.py tests/test_gpu_range.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "log or loopback or sharded_handle" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q 2>&1 | tail -4
echo "### config 2: bench new vs head"
cp $L/liblsqrhip.so /tmp/new.so
for r in 1 2; do
  cp /tmp/new.so $L/liblsqrhip.so
  for K in 2000 20; do echo "new K=$K"; timeout 300 python bench.py --steps $K --warmup $((K/10)) --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'], d['kernels']['spmv_mode2']['avg_launch_us'])"; done
  cp $L/liblsqrhip_head.so $L/liblsqrhip.so
  for K in 2000 20; do echo "head K=$K"; timeout 300 python bench.py --steps $K --warmup $((K/10)) --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'], d['kernels']['spmv_mode2']['avg_launch_us'])"; done
done
cp /tmp/new.so $L/liblsqrhip.so
echo "### pacing"
C4=random:10000000:10000000:100
for P in 0 2 4 8 16 64; do LSQRHIP_CSB_PACE=$P timeout 300 python scripts/kernel_times.py $C4 10; done
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum FETCH_SIZE"
for P in 4 16; do LSQRHIP_CSB_PACE=$P timeout 600 bash scripts/pmc_csb.sh $C4 pmc_c4_pace$P; done
} > $O 2>&1
tail -5 $O
