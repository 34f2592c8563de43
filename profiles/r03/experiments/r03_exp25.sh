#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
echo "### full gpu suite"
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
echo "### smoke"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "### driver-shaped bench"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_k20_new.json 2> gpurun_out/bench_k20_new.err
tail -3 gpurun_out/bench_k20_new.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_k20_new.json').read().strip().splitlines()[-1])
print('value', d['value'], 'roofline', {k:d['roofline'][k] for k in ('kernel','frac','achieved','traffic','bytes_per_launch','avg_launch_us')})
for h in d.get('roofline_hbm',[]): print(h.get('variant'), h.get('value'), h.get('roofline',{}).get('frac'), h.get('roofline',{}).get('avg_launch_us'), h.get('error'))
print('C4', d['strong_scaling_n1']['value'], d['strong_scaling_n1']['roofline']['frac'])
print('cpu', d['cpu_baseline'])
PY
echo "### fuzz"
timeout 900 python scripts/fuzz_layouts.py 60 31 2>&1 | tail -4
} > gpurun_out/r03_exp25.txt 2>&1
