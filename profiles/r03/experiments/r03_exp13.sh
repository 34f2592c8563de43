#!/bin/bash
# new engine tests, a wider fuzz, and the timeline of a 20-iteration solve on the round-3 build
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "### new tests"
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x -k "refuse or selected_device or more_gpus" 2>&1 | tail -5
echo "### fuzz (seeds 11, 12, 13; 80 cases each)"
for seed in 11 12 13; do timeout 900 python scripts/fuzz_layouts.py 80 $seed 2>&1 | tail -4; done
echo "### K=20 timeline"
rm -rf /tmp/tl && rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl -o t -- python3 scripts/short_solve_timeline.py 2>&1 | grep "wall us"
python3 scripts/short_solve_timeline.py --parse /tmp/tl 2>&1 | head -80
echo "### un-profiled"
python3 scripts/short_solve_timeline.py 2>&1 | grep "wall us"
python3 scripts/k20_eager_probe.py 2>&1 | head -3
} > gpurun_out/r03_exp13.txt 2>&1
