#!/bin/bash
# round 5: the kernels and HIP calls of one 20-iteration solve at config 2
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
rm -rf /tmp/k20tl; rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/k20tl -o t -- python3 scripts/short_solve_timeline.py > gpurun_out/r05/k20_timeline.txt 2>&1
python3 scripts/short_solve_timeline.py --parse /tmp/k20tl >> gpurun_out/r05/k20_timeline.txt 2>&1
python3 scripts/short_solve_timeline.py >> gpurun_out/r05/k20_timeline.txt 2>&1
