#!/bin/bash
# the round driver's command (20 steps after 5 warm-up days) several times + the two-rank plumbing hook
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-w}; mkdir -p $OUT; cd $R
for k in 1 2 3 4 5; do
  python bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=b['roofline']
print('ms/step %.6f kernels %.1f us frac %.5f cold %s' % (b['ms_per_step'], r['kernel_us_per_day'], r['frac'], b.get('cold_count_rows')), ' '.join('%s=%.1f' % (k,x['avg_launch_us']) for k,x in r['kernels'].items()))"
done
REINA_BENCH_BACKEND=gloo REINA_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-large --no-ensemble --no-cpu --no-sizes --strong-agents 4000000 > $OUT/${TAG}_bench2.json 2> $OUT/${TAG}_bench2.err
echo "two-rank hook rc=$?"; python -c "
import json
b=json.load(open('$OUT/${TAG}_bench2.json'))
print(b['n_gpus'], b['ms_per_step'], b.get('rccl_world'), list(b.keys()))
s=b.get('strong',{}); print('strong', s.get('workload'), s.get('ms_per_step'), s.get('rccl_world'), {k:x['avg_launch_us'] for k,x in s.get('roofline',{}).get('kernels',{}).items()})
"; tail -3 $OUT/${TAG}_bench2.err
