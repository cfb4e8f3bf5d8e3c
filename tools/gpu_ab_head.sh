#!/bin/bash
# this tree's library against the last commit's (built on the build host into reina_model_amd/csrc/variants/libreina_head.so), same box,
# same hour: per-day kernel times of the default year's first wave at 1e8 agents, the HUS year, the driver's window
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-abh}; mkdir -p $OUT; cd $R
lib() { if [ $1 = head ]; then export REINA_HIP_LIB=$R/reina_model_amd/csrc/variants/libreina_head.so; else unset REINA_HIP_LIB; fi; }
window() { python bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('$1 window ms/step %.6f kernels %.1f us' % (b['ms_per_step'], r['kernel_us_per_day']), ' '.join('%s=%.1f' % (k,x['avg_launch_us']) for k,x in r['kernels'].items()))"; }
{
for which in head new head new; do lib $which
  timeout 900 python tools/day_modes.py ${2:-100000000} ${3:-130} auto 2>&1 | grep -E "^# (mean|max)" | sed "s/^/$which ${2:-100000000} /"
done
for which in head new head new; do lib $which
  timeout 900 python tools/day_modes.py 1685983 365 auto 2>&1 | grep -E "^# (mean|max)" | sed "s/^/$which HUS /"
done
for which in head new head new head new; do lib $which; window $which; done
} | tee $OUT/${TAG}_ab.txt
if [ "${4:-parity}" = parity ]; then unset REINA_HIP_LIB
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "not hundred_million and not more_bed_events and not config2 and not config3 and not conservation and not two_hundred and not north_star" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log
fi
