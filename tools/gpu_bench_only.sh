#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; shift
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python bench.py --no-cpu "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"; tail -c 300 $OUT/${TAG}_bench.err
python - <<PY
import json
b=json.load(open('$OUT/${TAG}_bench.json'))
def show(r, name, v):
    print('%-10s %.4g agent-days/s  ms/step %.6f frac %.4f kernels %.1f us :' % (name, v, r['ms_per_step'], r['frac'], r['kernel_us_per_day']),
          ' '.join('%s=%.1f' % (k, x['avg_launch_us']) for k, x in r['kernels'].items()))
show(b['roofline'], 'headline', b['value'])
for k, v in b.get('full_scenario', {}).items():
    if 'error' in v: print(k, v)
    else: show(v['roofline'], k, v['value'])
e=b.get('ensemble', {})
print('ensemble', e.get('value'), e.get('ms_per_step'), e.get('kernels'), e.get('error'))
PY
