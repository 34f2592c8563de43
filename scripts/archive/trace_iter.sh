#!/bin/bash
# kernel timeline of the iteration loop under rocprofv3: trace_iter.sh PIPELINE [SPEC]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PIPE=${1:-1}; SPEC=${2:-poisson2d:1000:1000}
OUT=$R/gpurun_out/trace_pipe$PIPE
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/scripts/one_solve.py $SPEC 200 $PIPE > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
names = lambda n: n.split('(')[0].replace('void lsqrhip::','').replace('lsqrhip::','')[:28]
# steady-state window: last 300 kernels
w = rows[-320:-20]
agg = collections.defaultdict(list)
for s,e,n in w: agg[names(n)].append((e-s)/1e3)
print("pipeline=$PIPE  kernel averages over a steady window:")
for k,v in agg.items(): print(f"  {k:30s} n={len(v):4d} mean {sum(v)/len(v):7.2f} us  min {min(v):6.2f} max {max(v):6.2f}")
prev=None
print("timeline sample:")
for s,e,n in w[60:75]:
    print(f"  {names(n):30s} dur {(e-s)/1e3:6.2f} us   gap {((s-prev)/1e3 if prev else 0):6.2f}")
    prev=e
span = (w[-1][1]-w[0][0])/1e3
print("window span %.1f us for %d kernels" % (span, len(w)))
PY
