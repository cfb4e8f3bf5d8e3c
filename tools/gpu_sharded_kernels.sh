#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F=$OUT/sharded_day_kernels.txt; : > $F
for mode in exact mirror; do
echo "== 8 shards x 1685983 (BASELINE configs[1] per GPU), days 92-104 and 300-312 [$mode]" >> $F
python tools/sharded_kernels.py 8 13487864 92:104 $mode >> $F 2>&1
python tools/sharded_kernels.py 8 13487864 300:312 $mode >> $F 2>&1
echo "== 8 shards x 12.5e6 = 1e8 in total (north_star's target), days 92-104 and 300-312 [$mode]" >> $F
python tools/sharded_kernels.py 8 1e8 92:104 $mode >> $F 2>&1
python tools/sharded_kernels.py 8 1e8 300:312 $mode >> $F 2>&1
echo "== 2 shards x 5e7, days 92-104 and 300-312 [$mode]" >> $F
python tools/sharded_kernels.py 2 1e8 92:104 $mode >> $F 2>&1
python tools/sharded_kernels.py 2 1e8 300:312 $mode >> $F 2>&1
done
cat $F
