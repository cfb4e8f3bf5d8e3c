#!/bin/bash
# reina_read_history by a kernel writing the caller's page-locked block against the two copies (REINA_EXPORT=memcpy): where a 20-day
# run's wall time goes (tools/run20_breakdown.py), then the driver's window five times each
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-exp}; mkdir -p $OUT; cd $R
{
for m in kernel memcpy kernel memcpy; do
  if [ $m = memcpy ]; then export REINA_EXPORT=memcpy; else unset REINA_EXPORT; fi
  echo "== $m"; python tools/run20_breakdown.py 2>/dev/null
done
for i in 1 2 3 4 5; do for m in kernel memcpy; do
  if [ $m = memcpy ]; then export REINA_EXPORT=memcpy; else unset REINA_EXPORT; fi
  python bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', b['ms_per_step'], b['roofline']['kernel_us_per_day'])"
done; done
} | tee $OUT/${TAG}_export.txt
unset REINA_EXPORT
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "driver_contract or golden or kitchen" 2>&1 | tail -3
