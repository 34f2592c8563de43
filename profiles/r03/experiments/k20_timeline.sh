#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl -o t -- python3 scripts/short_solve_timeline.py 2>&1 | grep "wall us" > gpurun_out/k20_timeline_final.txt
python3 scripts/short_solve_timeline.py --parse /tmp/tl 2>&1 | head -60 >> gpurun_out/k20_timeline_final.txt
python3 scripts/short_solve_timeline.py 2>&1 | grep "wall us" >> gpurun_out/k20_timeline_final.txt
