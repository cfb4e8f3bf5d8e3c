#!/bin/bash
# round 4, first GPU call: the new sparse/dense tests + a slice of the suite, then per-day kernel times by mode
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "sparse or mini or ragged or random_scenarios or hus_full or group_equals or two_shards or initial_population" --durations=8 > $OUT/r4a_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/r4a_pytest.log
tail -25 $OUT/r4a_pytest.log
timeout 900 python tools/day_modes.py 100000000 365 dense sparse hotgather > $OUT/r4a_modes_100m.txt 2> $OUT/r4a_modes_100m.err
echo "modes rc=$?"; tail -5 $OUT/r4a_modes_100m.txt; tail -3 $OUT/r4a_modes_100m.err
timeout 300 python tools/day_modes.py 1685983 365 dense sparse > $OUT/r4a_modes_hus.txt 2> $OUT/r4a_modes_hus.err
echo "modes hus rc=$?"; tail -3 $OUT/r4a_modes_hus.txt
