#!/bin/bash
# randomised soak of the final binary: HIP vs oracle B bit for bit, far more scenarios than the suite holds
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout ${1:-900} python tools/parity_soak.py 100000 > $OUT/soak_unsharded.txt 2>&1; tail -3 $OUT/soak_unsharded.txt
timeout ${2:-600} python tools/parity_soak.py 100000 sharded > $OUT/soak_sharded.txt 2>&1; tail -3 $OUT/soak_sharded.txt
grep -c MISMATCH $OUT/soak_unsharded.txt $OUT/soak_sharded.txt
