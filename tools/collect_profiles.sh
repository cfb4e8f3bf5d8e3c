#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel traces + PMC traffic passes.
# Usage: bash tools/collect_profiles.sh <tag>      (writes gpurun_out/<tag>_*)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
for cfg in "hus:" "50m:--agents 50000000" "200m:--agents 200000000"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_$name -- python3 $R/bench.py --no-cpu --no-large --no-ensemble $args > $OUT/${TAG}_trace_$name.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch_$name -- python3 $R/bench.py --no-cpu --no-large --no-ensemble $args > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write_$name -- python3 $R/bench.py --no-cpu --no-large --no-ensemble $args > /dev/null 2>&1
done
echo collected
