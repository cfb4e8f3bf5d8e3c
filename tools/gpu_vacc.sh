#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-v}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu --durations=8 -k "vaccination or config2 or two_hundred or stepped_twice" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -16 $OUT/${TAG}_pytest.log
timeout 300 python tools/vacc_probe.py > $OUT/${TAG}_vacc_probe.txt 2>&1; tail -4 $OUT/${TAG}_vacc_probe.txt
timeout 300 python tools/vacc_probe.py 1e8 3333333 tiers > $OUT/${TAG}_vacc_probe_tiers.txt 2>&1; tail -4 $OUT/${TAG}_vacc_probe_tiers.txt
