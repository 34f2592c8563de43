#!/bin/bash
# column-swept products: the stream in 16-byte requests against four 4- and 8-byte requests each way; every shape in one
# process.  (The knob LSQRHIP_CSB_VEC existed in the builds this was run on -- profiles/r05/csb_vec_ab.txt -- and was not
# kept: slower on every shape.)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
{
for spec in random:10000000:10000000:100 random:1250000:10000000:100 random:4000000:1000000:100 powerlaw:5000000:2000000:10000 random:1250000:10000000:1000; do
  python scripts/ab_env.py $spec LSQRHIP_CSB_VEC=0,1 10 5
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/csb_vec_ab.txt
