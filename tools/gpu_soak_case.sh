#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for c in "$@"; do timeout 300 python tools/soak_case.py $c 2>&1 | grep -v amdgpu.ids | tail -25; done | tee $OUT/soak_cases.txt
