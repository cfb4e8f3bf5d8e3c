#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_xcd tools/ubench_xcd.hip && timeout 300 /tmp/ubench_xcd > $OUT/ubench_xcd.txt 2>&1
echo "rc=$?"; cat $OUT/ubench_xcd.txt
