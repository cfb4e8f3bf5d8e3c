#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-sq}; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "two_shards or random_scenarios_on or north_star" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log
F=$OUT/${TAG}_kernels.txt; : > $F
for mode in exact mirror; do
python tools/sharded_kernels.py 8 1e8 92:104 $mode 2>/dev/null >> $F
python tools/sharded_kernels.py 2 1e8 92:104 $mode 2>/dev/null >> $F
done
cat $F
