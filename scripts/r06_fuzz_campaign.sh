#!/bin/bash
# r06: more seeds of the layout fuzz on the FINAL build than the suite runs (the fused column splits are layouts 14-16; the
# engine variants force them on small blocks), plus round 5's failing seed 605 under the new rule
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
bash scripts/fuzz_campaign.sh gpurun_out/r06/fuzz_campaign_seeds711_714.txt 711 712 713 714
bash scripts/fuzz_campaign.sh gpurun_out/r06/fuzz_campaign_seed605.txt 605
FUZZ_ARGS=--engine bash scripts/fuzz_campaign.sh gpurun_out/r06/fuzz_campaign_engine_seeds721_722.txt 721 722
FUZZ_ARGS=--real32 bash scripts/fuzz_campaign.sh gpurun_out/r06/fuzz_campaign_real32_seed731.txt 731
grep -h "failures" gpurun_out/r06/fuzz_campaign_*.txt
