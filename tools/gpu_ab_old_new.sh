#!/bin/bash
# A/B on ONE box: the round-4 tree (old_tree_r04.tar, unpacked and built in /tmp) against this tree -- the driver's window and
# per-kernel times of the HUS year and of 1e8 agents
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-ab}; mkdir -p $OUT
cd /tmp && rm -rf old && tar xf $R/old_tree_r04.tar && cd /tmp/old && python -c "from reina_model_amd import build; build.build(verbose=False)" > /dev/null 2>&1
window() { python bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('$1 ms/step %.6f kernels %.1f us' % (b['ms_per_step'], r['kernel_us_per_day']), ' '.join('%s=%.1f' % (k,x['avg_launch_us']) for k,x in r['kernels'].items()))"; }
for k in 1 2 3; do (cd /tmp/old && window old); (cd $R && window new); done | tee $OUT/${TAG}_window.txt
for n in 100000000; do
 (cd /tmp/old && python tools/day_modes.py $n 365 auto 2>/dev/null | grep "^# mean" | sed 's/^/old /'); (cd $R && python tools/day_modes.py $n 365 auto 2>/dev/null | grep "^# mean" | sed 's/^/new /')
done | tee $OUT/${TAG}_modes.txt
