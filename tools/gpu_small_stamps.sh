#!/bin/bash
# in-kernel stamps of k_small_day (-DREINA_SMALL_STAMPS): bash tools/gpu_small_stamps.sh "<wgs> ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DREINA_SMALL_STAMPS -o /tmp/libreina_ss.so reina_model_amd/csrc/reina_hip.hip 2>&1 | grep error
for w in ${1:-32 16}; do echo "== $w workgroups"; REINA_FUSED_WGS=$w REINA_HIP_LIB=/tmp/libreina_ss.so python tools/small_stamps.py 2>/dev/null; done | tee $OUT/stamps_small.txt
