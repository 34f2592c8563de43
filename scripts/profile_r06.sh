#!/bin/bash
# Round-6 evidence (same recipe as rounds 4 and 5): one `rocprofv3 --kernel-trace --stats` summary per reported workload, from ONE build,
# next to the bench.py line of the same command (profiles/r06/).  PMC traffic comes from bench.py itself
# (its --traffic live passes); under rocprofv3 bench.py skips them.  usage: scripts/profile_r06.sh [tags...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06f
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
setenvs() { if [ "$1" != "-" ]; then for e in ${1//,/ }; do export $e; done; fi; }     # "A=1,B=2" or "-"
unsetenvs() { if [ "$1" != "-" ]; then for e in ${1//,/ }; do unset ${e%%=*}; done; fi; }
run() {   # tag, env assignments ("A=1,B=2") or "-", bench args...
  local tag=$1 envs=$2; shift 2
  local d=$OUT/$tag
  rm -rf "$d"; mkdir -p "$d"
  setenvs "$envs"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -o t -- python3 "$R/bench.py" "$@" --traffic off --cpu-iters 0 --extras off --detail "$d/detail.json" > "$d/bench.json" 2> "$d/err.txt"
  unsetenvs "$envs"
  local f=$(find "$d/trace" -name "*kernel_stats.csv" | head -1)
  {
    echo "# $tag: rocprofv3 --kernel-trace --stats -- python3 bench.py $* --traffic off --cpu-iters 0   ${envs}"
    echo "# bench.py line of the same run:"
    tail -1 "$d/bench.json" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d.get('roofline',{})
print('#   value %.1f it/s  ms_per_step %.4f' % (d['value'], d['ms_per_step']))
print('#   roofline (SURVEY 8d bytes): kernel %s | B1 %d | avg_launch_us %.2f (x%d kernel launches: %.2f us each) | achieved %.0f GB/s | frac %.3f | frac_mode2 %.3f (%.2f us) | frac_layout %.3f' % (r.get('kernel'), r.get('bytes_per_launch',0), r.get('avg_launch_us',0), r.get('kernel_launches_per_product',1), r.get('avg_kernel_launch_us',0), r.get('achieved',0), r.get('frac',0), r.get('frac_mode2',0), r.get('avg_launch_us_mode2',0), r.get('frac_layout',0)))
" 2>/dev/null
    echo "# kernel stats (Name, Calls, TotalDurationNs, AverageNs, Percentage):"
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-96s calls=%7s total_ns=%13s avg_ns=%11s pct=%6s" % (r.get("Name", "")[:96], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
PY
  } > "$OUT/$tag.txt"
  tail -1 "$d/bench.json" > "$OUT/$tag.bench.json"
  cat "$OUT/$tag.txt"
}
# the dominant kernel alone: its rocprofv3 average x launches per product = roofline.avg_launch_us
roof() {   # tag, env assignment or "-", workload [, mode]
  local tag=$1 envs=$2 spec=$3 mode=${4:-1}
  [ "$mode" = 2 ] && tag=${tag}_mode2
  local d=$OUT/${tag}_roofline
  rm -rf "$d"; mkdir -p "$d"
  setenvs "$envs"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -o t -- python3 "$R/bench.py" --workload $spec --roofline-only --roofline-mode $mode > "$d/bench.json" 2> "$d/err.txt"
  unsetenvs "$envs"
  local f=$(find "$d/trace" -name "*kernel_stats.csv" | head -1)
  {
    echo "# ${tag}_roofline: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $spec --roofline-only --roofline-mode $mode   ${envs}"
    echo "# bench.py line of the same run (HIP events on the solver's stream):"
    echo "#   $(tail -1 "$d/bench.json")"
    echo "# kernel stats of the product kernels (Name, Calls, TotalDurationNs, AverageNs):"
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r.get("Name", "") for k in ("k_spmv_", "k_panel_combine", "k_csb_combine", "k_csb_xmax")):
        print("%-110s calls=%6s total_ns=%13s avg_ns=%12s" % (r["Name"][:110], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs")))
PY
  } > "$OUT/${tag}_roofline.txt"
  cat "$OUT/${tag}_roofline.txt"
}
TAGS=${*:-"config2 config2_packed poisson4000_pat poisson4000_dict poisson4000_val8 poisson4000_spat config4 shard8 shard8_r1000 shard8_plan config3_100 config5"}
for t in $TAGS; do
  case $t in
    config2)          run config2_poisson1000 - --workload poisson2d:1000:1000 --extras off --steps 2000 --warmup 200
                      roof config2_poisson1000 - poisson2d:1000:1000 ;;
    config2_packed)   run config2_packed_records LSQRHIP_PAT=0 --workload poisson2d:1000:1000 --extras off --steps 2000 --warmup 200
                      roof config2_packed_records LSQRHIP_PAT=0 poisson2d:1000:1000 ;;
    poisson4000_pat)  run poisson4000_patterns - --workload poisson2d:4000:4000 --steps 200 --warmup 20
                      roof poisson4000_patterns - poisson2d:4000:4000 ;;
    poisson4000_dict) run poisson4000_dict LSQRHIP_PAT=0 --workload poisson2d:4000:4000 --steps 200 --warmup 20
                      roof poisson4000_dict LSQRHIP_PAT=0 poisson2d:4000:4000 ;;
    poisson4000_val8) run poisson4000_val8 LSQRHIP_PAT=0,LSQRHIP_VAL8=0,LSQRHIP_SPAT=0 --workload poisson2d:4000:4000 --steps 200 --warmup 20
                      roof poisson4000_val8 LSQRHIP_PAT=0,LSQRHIP_VAL8=0,LSQRHIP_SPAT=0 poisson2d:4000:4000 ;;
    poisson4000_spat) run poisson4000_structure_patterns LSQRHIP_PAT=0,LSQRHIP_VAL8=0 --workload poisson2d:4000:4000 --steps 200 --warmup 20
                      roof poisson4000_structure_patterns LSQRHIP_PAT=0,LSQRHIP_VAL8=0 poisson2d:4000:4000 ;;
    mesh4000_wide)    run mesh4000_wide_patterns - --workload mesh2d:4000:4000:16:16 --steps 200 --warmup 20
                      roof mesh4000_wide_patterns - mesh2d:4000:4000:16:16 ;;
    config4)          run config4_random_10Mx10Mx100 - --workload random:10000000:10000000:100 --steps 20 --warmup 2
                      roof config4_random_10Mx10Mx100 - random:10000000:10000000:100
                      roof config4_random_10Mx10Mx100 - random:10000000:10000000:100 2 ;;
    shard8)           run shard8_random_1250000x10Mx100 - --workload random:1250000:10000000:100 --steps 40 --warmup 4
                      roof shard8_random_1250000x10Mx100 - random:1250000:10000000:100
                      roof shard8_random_1250000x10Mx100 - random:1250000:10000000:100 2 ;;
    config3_100)      run config3_random_4Mx1Mx100 - --workload random:4000000:1000000:100 --steps 40 --warmup 4
                      roof config3_random_4Mx1Mx100 - random:4000000:1000000:100 ;;
    config3_literal)  run config3_literal_4Mx1Mx1000 - --workload random:4000000:1000000:1000 --steps 10 --warmup 2
                      roof config3_literal_4Mx1Mx1000 - random:4000000:1000000:1000 ;;
    config5)          run config5_powerlaw_5Mx2M - --workload powerlaw:5000000:2000000:10000 --steps 40 --warmup 4
                      roof config5_powerlaw_5Mx2M - powerlaw:5000000:2000000:10000
                      roof config5_powerlaw_5Mx2M - powerlaw:5000000:2000000:10000 2 ;;
    shard8_r1000)     run shard8_random_1250000x10Mx1000 - --workload random:1250000:10000000:1000 --steps 10 --warmup 2
                      roof shard8_random_1250000x10Mx1000 - random:1250000:10000000:1000 ;;
    shard8_plan)      run shard8_overlap_plan LSQRHIP_SHARD_OVERLAP=1,LSQRHIP_SHARD_WORLD=8 --workload random:1250000:10000000:100 --steps 40 --warmup 4
                      roof shard8_overlap_plan LSQRHIP_SHARD_OVERLAP=1,LSQRHIP_SHARD_WORLD=8 random:1250000:10000000:100 ;;
  esac
done
