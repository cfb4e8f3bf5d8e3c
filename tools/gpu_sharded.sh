#!/bin/bash
# the sharded parity tests (both attribution modes) + the RCCL world-1 paths
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-s}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_distributed.py -x -q -m gpu --durations=10 -k "shard or nccl or north_star or smoke" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -30 $OUT/${TAG}_pytest.log
