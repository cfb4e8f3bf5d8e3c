#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-t}; shift; mkdir -p $OUT; cd $R
timeout 1800 python -m pytest tests -q -m gpu "$@" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/${TAG}_pytest.log | tail -40
