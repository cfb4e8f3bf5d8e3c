#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
bash tools/gpu_call.sh d
cd /tmp && export TMPDIR=/tmp
for cfg in "100m:--agents 100000000" "hus:"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d_trace_$name -- python3 $R/bench.py --no-cpu --no-sizes --no-ensemble --steps 365 --warmup 0 --preheat-days 0 $args > $OUT/d_trace_$name.json 2>/dev/null
  echo "trace $name rc=$?"
done
