#!/bin/bash
# more seeds of the layout fuzz than the test suite runs (tests/fuzz_layouts.py): usage [FUZZ_ARGS=--engine] fuzz_campaign.sh OUT seed...
cd ${GRAFT_REPO_ROOT:-.}
OUT=$1; shift
for seed in "$@"; do
  echo "== seed $seed" >> $OUT
  timeout 900 python tests/fuzz_layouts.py 150 $seed $FUZZ_ARGS 2>&1 | tail -4 >> $OUT
done
