#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_exp8.txt
{
python scripts/fuzz_case.py 1 4
LSQRHIP_LIB=liblsqrhip_head.so python scripts/fuzz_case.py 1 4
echo "### fuzz seeds"
for s in 1 2 3; do timeout 900 python scripts/fuzz_layouts.py 40 $s 2>&1 | cut -c1-200 | tail -8; done
echo "### rest of the suite"
timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fuzz.py 2>&1 | tail -15
} > $O 2>&1
tail -5 $O
