#!/bin/bash
# quick GPU check: the parity tests that run in seconds + per-day kernel times of the HUS year and of 10^8 agents
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-q}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "not hundred_million and not more_bed_events and not config2 and not config3 and not conservation and not north_star" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -5 $OUT/${TAG}_pytest.log
timeout 300 python tools/day_modes.py 1685983 365 auto > $OUT/${TAG}_modes_hus.txt 2>/dev/null; grep "^# mean" $OUT/${TAG}_modes_hus.txt
awk 'NR>2 && NR%30==27' $OUT/${TAG}_modes_hus.txt
timeout 300 python tools/day_modes.py 100000000 365 auto > $OUT/${TAG}_modes_100m.txt 2>/dev/null; grep "^# mean" $OUT/${TAG}_modes_100m.txt
awk 'NR>2 && NR%24==0' $OUT/${TAG}_modes_100m.txt
