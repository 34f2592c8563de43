#!/bin/bash
# r06, first GPU call: the new bench line (driver form K = 20 and the default), the tests that are new this round
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06
mkdir -p $OUT
( time python bench.py --steps 20 --warmup 5 --detail $OUT/bench_detail_k20.json ) > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
tail -c 4200 $OUT/bench_default_k20.json; echo; tail -5 $OUT/bench_default_k20.err
python -m pytest tests/test_gpu_devgen.py -q -k "complete_when_solve_returns" 2>&1 | tail -3
python -m pytest tests/test_gpu_patterns.py -q -k "paired_rows_and_the_slice" 2>&1 | tail -3
python -m pytest tests/test_gpu_engine.py -q -k "world_one_and_its_fallback or python_stage_driver_over" 2>&1 | tail -3
