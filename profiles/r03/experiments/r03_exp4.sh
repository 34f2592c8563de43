#!/bin/bash
# r03 experiment 4: drift trace of the sweeps; config 2 in-solve kernel times for three builds; engine tests
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
L=lsqr_amd/lib
O=$R/gpurun_out/r03_exp4.txt
mkdir -p gpurun_out
{
echo "### engine tests"
timeout 1500 python -m pytest tests/test_gpu_engine.py -q 2>&1 | tail -40
echo "### drift trace"
C4=random:10000000:10000000:100
LSQRHIP_LIB=liblsqrhip_trace.so timeout 300 python scripts/csb_trace.py $C4
LSQRHIP_LIB=liblsqrhip_trace.so LSQRHIP_CSB_PACE=4 timeout 300 python scripts/csb_trace.py $C4
echo "### config 2 in-solve"
cd /tmp && export TMPDIR=/tmp
for lib in liblsqrhip_head.so liblsqrhip_early0.so liblsqrhip.so; do
  for r in 1 2; do
    echo "== $lib K=2000 run $r"
    LSQRHIP_LIB=$lib timeout 300 python $R/bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'], d['kernels']['spmv_mode2']['avg_launch_us'])"
  done
  rm -rf /tmp/prof_$lib
  LSQRHIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -o p -- python3 $R/bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 --no-roofline > /dev/null 2>&1
  F=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib kernel stats"; head -8 "$F" | cut -c1-200
done
} > $O 2>&1
tail -5 $O
