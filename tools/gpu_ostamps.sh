#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DREINA_OPEN_STAMPS -o /tmp/libreina_os.so reina_model_amd/csrc/reina_hip.hip 2>/dev/null
REINA_HIP_LIB=/tmp/libreina_os.so python tools/open_stamps_big.py 1e8 2>&1 | tee $OUT/stamps_open_1e8.txt
REINA_HIP_LIB=/tmp/libreina_os.so python tools/open_stamps_big.py 1685983 2>&1 | tee $OUT/stamps_open_hus.txt
