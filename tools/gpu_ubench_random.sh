#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_random tools/ubench_random.hip 2>/dev/null && timeout 600 /tmp/ubench_random > $OUT/ubench_random.txt 2>&1
echo "rc=$?"; cat $OUT/ubench_random.txt
