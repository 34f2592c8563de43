#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_us'],2))"; }
{
for r in 1 2; do
LSQRHIP_LIB=liblsqrhip.so timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=2 grid=1024"
LSQRHIP_LIB=liblsqrhip_u4.so timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=4 grid=1024"
LSQRHIP_LIB=liblsqrhip_u3.so LSQRHIP_PAT_GRID=1280 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=3 grid=1280"
LSQRHIP_LIB=liblsqrhip_u3.so LSQRHIP_PAT_GRID=1024 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=3 grid=1024"
LSQRHIP_LIB=liblsqrhip_u4.so LSQRHIP_PAT_GRID=768 timeout 300 python bench.py --steps 2000 --warmup 200 --extras off --traffic off --cpu-iters 0 2>/dev/null | line "U=4 grid=768"
done
} > gpurun_out/r03_exp38.txt 2>&1
