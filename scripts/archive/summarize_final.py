#!/usr/bin/env python3
"""One line per bench line under a directory of scripts/r04_final.sh outputs (default gpurun_out/r04)."""
import glob, json, os, sys
d0 = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04"
def last(f):
    try:
        return json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    except Exception:
        return None
for f in ['bench_default.json', 'bench_default_k20.json', 'engine_1rank_shard8.json', 'engine_py_1rank_shard8.json', 'engine_eager_1rank_shard8.json']:
    d = last(os.path.join(d0, f))
    if not d:
        print(f, 'NONE'); continue
    r = d['roofline']
    print(f, round(d['value'], 1), round(d['ms_per_step'], 5), 'frac', round(r['frac'], 3), 'of_ceiling', r.get('of_ceiling'), 'traffic', r.get('traffic'), r.get('bytes_per_launch'), r.get('avg_launch_us'))
    if 'strong_scaling_n1' in d:
        n1 = d['strong_scaling_n1']; q = n1['roofline']
        print('   n1', n1.get('value'), q['frac'], q.get('of_ceiling'), q.get('traffic'), q.get('bytes_per_launch'), q.get('avg_launch_us'), q.get('kernel_launches_per_product'))
        print('   general', d['roofline_general']['frac'], [(e.get('variant')[:14], round(e['roofline']['frac'], 3)) for e in d['roofline_hbm']], d['cpu_baseline']['value'])
for f in sorted(glob.glob(os.path.join(d0, '*.bench.json'))):
    d = last(f)
    if d:
        k = d.get('kernels', {}).get('spmv_mode2', {})
        print(os.path.basename(f), round(d['value'], 1), 'frac', round(d['roofline']['frac'], 3), d['roofline'].get('of_ceiling'), 'avg_launch_us', round(d['roofline'].get('avg_launch_us', 0), 1), 'm2', round(k.get('frac', 0), 3), round(k.get('avg_launch_us', 0), 1))
