#!/bin/bash
# rocprofv3 kernel trace of one 365-day scenario: bash tools/gpu_trace.sh <tag> [agents] ; summary printed by prof_summary.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-p}; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for n in ${2:-100000000}; do
A="--no-cpu --no-sizes --no-ensemble --steps 365 --warmup 0 --preheat-days 0 --agents $n"
rm -rf $OUT/${TAG}_${n}_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${n}_trace -- python3 $R/bench.py $A > $OUT/${TAG}_${n}_trace.json 2>/dev/null
echo "trace $n rc=$?"
(cd $R && python tools/prof_summary.py ${TAG}_${n})
# keep only the small csvs
find $OUT/${TAG}_${n}_trace -name "*kernel_trace.csv" -size +20M -delete
done
