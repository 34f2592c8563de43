#!/bin/bash
# the multi-process tests three times over (one failure in five full runs of the suite could not be named: its output was filtered)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
OUT=gpurun_out/r06/flaky_hunt.txt
: > $OUT
for i in 1 2 3; do
  python -m pytest tests/test_dist.py tests/test_gpu_engine.py tests/test_fortran.py tests/test_gpu_devgen.py -m gpu -q -rf 2>&1 | tail -25 >> $OUT
done
grep -E "passed|failed|FAILED" $OUT
