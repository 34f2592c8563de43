#!/bin/bash
# wide row patterns: the pattern / parity / format / fuzz tests, the timing table, config 2 at K = 20 and 2000 (unchanged?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" && mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_parity.py tests/test_gpu_formats.py tests/test_gpu_fuzz.py tests/test_gpu_real32.py -x -q -m gpu 2>&1 | tail -4
timeout 500 python scripts/r05_wide_patterns.py 2>&1 | tee gpurun_out/r05/wide_patterns.txt
for K in 20 2000 20 2000; do
  timeout 300 python bench.py --steps $K --warmup 5 --extras off --traffic off --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2  K =', d['steps'], ':', round(d['value'],1), 'it/s')"
done
