#!/bin/bash
# in-kernel stamps of k_day (-DREINA_DAY_STAMPS): bash tools/gpu_dstamps.sh "<agents> ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DREINA_DAY_STAMPS -o /tmp/libreina_ds.so reina_model_amd/csrc/reina_hip.hip 2>&1 | grep error
for n in ${1:-1685983 100000000}; do echo "== $n"; REINA_HIP_LIB=/tmp/libreina_ds.so python tools/day_stamps.py $n 2>/dev/null; done | tee $OUT/stamps_day.txt
