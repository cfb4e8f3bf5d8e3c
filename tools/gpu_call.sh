#!/bin/bash
# one GPU call of the build -> measure loop: GPU suite, bench line, optionally kernel traces
# usage: bash tools/gpu_call.sh <tag> [pytest-args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu --durations=12 "$@" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
tail -32 $OUT/${TAG}_pytest.log
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"; tail -c 300 $OUT/${TAG}_bench.err
python - <<PY
import json
try:
    b=json.load(open('$OUT/${TAG}_bench.json'))
except Exception as e:
    print('no bench json', e); raise SystemExit
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
# the N > 1 plumbing on a 1-GPU box: bench.py launches its own two ranks (gloo, both on cuda:0) -- not a measurement
REINA_BENCH_BACKEND=gloo REINA_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-large --no-ensemble --no-cpu --no-sizes > $OUT/${TAG}_bench2.json 2> $OUT/${TAG}_bench2.err
echo "two-rank hook rc=$?"; head -c 400 $OUT/${TAG}_bench2.json; echo
