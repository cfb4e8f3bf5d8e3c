#!/bin/bash
# round 4: full GPU suite, then per-day kernel times by mode at four sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-r4b}; mkdir -p $OUT; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu --durations=10 > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
tail -22 $OUT/${TAG}_pytest.log
for cfg in "100000000:dense sparse" "50000000:dense sparse" "200000000:dense sparse" "1685983:dense sparse"; do
  n=${cfg%%:*}; modes=${cfg#*:}
  timeout 900 python tools/day_modes.py $n 365 $modes > $OUT/${TAG}_modes_$n.txt 2> $OUT/${TAG}_modes_$n.err
  echo "modes $n rc=$?"; grep "^# mean" $OUT/${TAG}_modes_$n.txt; tail -2 $OUT/${TAG}_modes_$n.err
done
