#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-p}; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
A="--no-cpu --no-sizes --no-ensemble --steps 365 --warmup 0 --preheat-days 0 --agents ${2:-100000000}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py $A > $OUT/${TAG}_trace.json 2>/dev/null
echo "trace rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_sq -- python3 $R/bench.py $A > /dev/null 2>&1
echo "sq rc=$?"
