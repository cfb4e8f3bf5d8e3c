#!/bin/bash
# the parity tests that run in seconds (all scenario families, sharded both ways) + sharded kernel costs
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-pq}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "not hundred_million and not more_bed_events and not config2 and not config3 and not conservation and not two_hundred" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -5 $OUT/${TAG}_pytest.log
F=$OUT/${TAG}_kernels.txt; : > $F
for mode in exact mirror; do
python tools/sharded_kernels.py 8 1e8 92:104 $mode 2>/dev/null >> $F
python tools/sharded_kernels.py 2 1e8 92:104 $mode 2>/dev/null >> $F
done
cat $F
